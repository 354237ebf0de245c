/*
 * vec_sim_debug.h -- the reference's debug entry points for HNSW graphs (src/VecSim/vec_sim_debug.h:30-44), same names and
 * argument meaning, over the host-built graph of libvecsim_amd.so's HNSW index.
 *
 * VecSimDebug_GetElementNeighborsInHNSWGraph: *neighborsData becomes an array of <top level + 2> entries; entry l
 * (0 <= l <= top level) is an int array {n_l, label_1 .. label_n_l} -- the labels of the element's neighbours on level l --
 * and the last entry is NULL.  Returns a VecSimDebugCommandCode: OK, BadIndex (not an HNSW index), LabelNotExists,
 * MultiNotSupported (multi-value indexes, as upstream).  Free with VecSimDebug_ReleaseElementNeighborsInHNSWGraph.
 */
#pragma once
#include "vec_sim.h"

#ifdef __cplusplus
extern "C" {
#endif

int VecSimDebug_GetElementNeighborsInHNSWGraph(VecSimIndex *index, size_t label, int ***neighborsData);
void VecSimDebug_ReleaseElementNeighborsInHNSWGraph(int **neighborsData);

#ifdef __cplusplus
}
#endif
