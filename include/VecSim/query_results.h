/*
 * VecSim/query_results.h -- reply / iterator accessors of the C boundary.
 * Same symbols and signatures as the reference's src/VecSim/query_results.h:21-138; the objects
 * behind the opaque pointers are ours (vectorsimilarity_amd/csrc/host/reply.cpp).
 */
#pragma once
#include <stdbool.h>
#include <stdlib.h>

#include "vec_sim_common.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { BY_SCORE, BY_ID, BY_SCORE_THEN_ID } VecSimQueryReply_Order;
typedef enum { VecSim_QueryReply_OK = VecSim_OK, VecSim_QueryReply_TimedOut } VecSimQueryReply_Code;

typedef struct VecSimQueryResult VecSimQueryResult;                 /* one (label, score) hit  */
typedef struct VecSimQueryReply VecSimQueryReply;                   /* caller-owned hit list   */
typedef struct VecSimQueryReply_Iterator VecSimQueryReply_Iterator; /* cursor over a reply     */
typedef struct VecSimBatchIterator VecSimBatchIterator;             /* "next n best" cursor    */

/* a NULL item yields INVALID_ID / NaN (query_results.cpp:54-66) */
int64_t VecSimQueryResult_GetId(const VecSimQueryResult *item);
double VecSimQueryResult_GetScore(const VecSimQueryResult *item);

size_t VecSimQueryReply_Len(VecSimQueryReply *results);
VecSimQueryReply_Code VecSimQueryReply_GetCode(VecSimQueryReply *results);
void VecSimQueryReply_Free(VecSimQueryReply *results);

VecSimQueryReply_Iterator *VecSimQueryReply_GetIterator(VecSimQueryReply *results);
VecSimQueryResult *VecSimQueryReply_IteratorNext(VecSimQueryReply_Iterator *iterator);
bool VecSimQueryReply_IteratorHasNext(VecSimQueryReply_Iterator *iterator);
void VecSimQueryReply_IteratorReset(VecSimQueryReply_Iterator *iterator);
void VecSimQueryReply_IteratorFree(VecSimQueryReply_Iterator *iterator);

VecSimQueryReply *VecSimBatchIterator_Next(VecSimBatchIterator *iterator, size_t n_results,
                                           VecSimQueryReply_Order order);
bool VecSimBatchIterator_HasNext(VecSimBatchIterator *iterator);
void VecSimBatchIterator_Free(VecSimBatchIterator *iterator);
void VecSimBatchIterator_Reset(VecSimBatchIterator *iterator);

#ifdef __cplusplus
}
#endif
