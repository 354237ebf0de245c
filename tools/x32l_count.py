import sys, os
sys.path.insert(0, "/root/repo")
from vectorsimilarity_amd import VecSim, synth
p = VecSim.BFParams()
p.type, p.dim, p.metric = VecSim.VecSimType_INT8, 1024, VecSim.VecSimMetric_Cosine
ix = VecSim.BFIndex(p)
ix.add_synthetic(int(sys.argv[2]), 42)
q = synth.rows_i8(48, 0, 256, 1024)
ix.set_option("lowp_x32", int(sys.argv[1]))
ix.knn_query(q, 100)
