/* Prints sizeof/offsetof of every public struct of the C boundary as JSON.  Compiled twice: by
 * make_abi_layout.py against the REFERENCE header (-> abi_layout.json, committed fixture) and by
 * tests/test_abi.py against include/VecSim/vec_sim_common.h (must print the same). */
#include <stddef.h>
#include <stdio.h>
#include VECSIM_COMMON_HEADER

#define S(T) printf("  \"sizeof(" #T ")\": %zu,\n", sizeof(T))
#define O(T, f) printf("  \"offsetof(" #T "," #f ")\": %zu,\n", offsetof(T, f))
int main(void) {
    printf("{\n");
    S(BFParams); O(BFParams, dim); O(BFParams, metric); O(BFParams, multi); O(BFParams, initialCapacity); O(BFParams, blockSize);
    S(HNSWParams); O(HNSWParams, blockSize); O(HNSWParams, M); O(HNSWParams, efConstruction); O(HNSWParams, efRuntime); O(HNSWParams, epsilon);
    S(SVSParams); O(SVSParams, quantBits); O(SVSParams, alpha); O(SVSParams, num_threads); O(SVSParams, epsilon);
    S(TieredIndexParams); O(TieredIndexParams, primaryIndexParams); O(TieredIndexParams, specificParams);
    S(AlgoParams);
    S(VecSimParams); O(VecSimParams, algoParams); O(VecSimParams, logCtx);
    S(VecSimQueryParams); O(VecSimQueryParams, batchSize); O(VecSimQueryParams, searchMode); O(VecSimQueryParams, timeoutCtx);
    S(VecSimRawParam);
    S(VecSimIndexBasicInfo); O(VecSimIndexBasicInfo, isMulti); O(VecSimIndexBasicInfo, blockSize); O(VecSimIndexBasicInfo, dim);
    S(VecSimIndexStatsInfo);
    S(CommonInfo); O(CommonInfo, indexSize); O(CommonInfo, memory); O(CommonInfo, lastMode);
    S(hnswInfoStruct); S(svsInfoStruct); S(tieredInfoStruct); O(tieredInfoStruct, bfInfo); O(tieredInfoStruct, bufferLimit);
    S(VecSimIndexDebugInfo);
    S(VecSimMemoryFunctions); S(VecSimDiskContext); S(VecSimParamsDisk);
    printf("  \"VecSimType_UINT8\": %d, \"VecSimAlgo_SVS\": %d, \"VecSimMetric_Cosine\": %d, \"RANGE_QUERY\": %d,\n",
           (int)VecSimType_UINT8, (int)VecSimAlgo_SVS, (int)VecSimMetric_Cosine, (int)RANGE_QUERY);
    printf("  \"VecSimParamResolverErr_InvalidPolicy_AdHoc_With_EfRuntime\": %d, \"QUERY_TYPE_RANGE\": %d,\n",
           (int)VecSimParamResolverErr_InvalidPolicy_AdHoc_With_EfRuntime, (int)QUERY_TYPE_RANGE);
    printf("  \"DEFAULT_BLOCK_SIZE\": %d\n}\n", DEFAULT_BLOCK_SIZE);
    return 0;
}
