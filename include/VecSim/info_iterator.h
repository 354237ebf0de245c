/*
 * info_iterator.h -- C-ABI restatement of the reference's debug-info iterator
 * (reference: src/VecSim/info_iterator.h:21-89).  Same type names, enum order, field struct layout and
 * function names, so a caller compiled against the reference header links unchanged.
 */
#pragma once
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>

#include "vec_sim_common.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct VecSimDebugInfoIterator VecSimDebugInfoIterator;

/* reference: info_iterator.h:23-29 */
typedef enum {
    INFOFIELD_STRING,
    INFOFIELD_INT64,
    INFOFIELD_UINT64,
    INFOFIELD_FLOAT64,
    INFOFIELD_ITERATOR
} VecSim_InfoFieldType;

/* reference: info_iterator.h:31-37 */
typedef union {
    double floatingPointValue;
    int64_t integerValue;
    uint64_t uintegerValue;
    const char *stringValue;
    VecSimDebugInfoIterator *iteratorValue;
} FieldValue;

/* reference: info_iterator.h:45-49 */
typedef struct {
    const char *fieldName;
    VecSim_InfoFieldType fieldType;
    FieldValue fieldValue;
} VecSim_InfoField;

/* reference: info_iterator.h:57, 66, 74, 81.  The iterator is created by VecSimIndex_DebugInfoIterator
 * (vec_sim.h) and owned by the caller until _Free. */
size_t VecSimDebugInfoIterator_NumberOfFields(VecSimDebugInfoIterator *infoIterator);
bool VecSimDebugInfoIterator_HasNextField(VecSimDebugInfoIterator *infoIterator);
VecSim_InfoField *VecSimDebugInfoIterator_NextField(VecSimDebugInfoIterator *infoIterator);
void VecSimDebugInfoIterator_Free(VecSimDebugInfoIterator *infoIterator);

#ifdef __cplusplus
}
#endif
