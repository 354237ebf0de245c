# VSGPU_VERIFY (dense re-check of every filter-path reply, vsgpu.hip) over multi-slab tables of wide rows and SQ8 rows: prints
# "VSGPU_VERIFY MISS ..." for every row the filter pipeline lost; expected output: the bench lines only
export VSGPU_VERIFY=1
for t in bf16 f16; do timeout 600 python tools/bench_dims.py --type $t 3072 4096 6144 8192 2>&1 | grep -c "MISS" ; done
timeout 600 python tools/bench_dims.py --type f32 4096 6144 8192 2>&1 | grep -c "MISS"
timeout 600 python tools/bench_dims.py --type bf16 --metric IP --batch 40 5000 2>&1 | grep -c MISS
timeout 900 python tools/bench_sq8.py --rows 3000000 --batches 64,128 --steps 2 --metric L2 2>&1 | grep -c MISS
timeout 900 python tools/bench_sq8.py --rows 3000000 --batches 128 --steps 2 --metric IP 2>&1 | grep -c MISS
