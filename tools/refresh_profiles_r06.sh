# usage (GPU box, repo root): bash tools/refresh_profiles_r06.sh -> everything profiles/r06_* holds that comes from the final build, written to gpurun_out/r06/
# (the four BASELINE configurations through tools/refresh_profiles_config.sh: PMC traffic passes, timed-replay kernel trace, bench line; then the
# small-mesh bench lines, the per-rank floors of the sharded forward, the 8-ranks-on-one-GPU run, the 1024-channel line and the SQ counter passes)
R=$PWD; OUT=$R/gpurun_out/r06; mkdir -p $OUT
for c in o96 o96-res6 n320 gnn; do ROUND=r06 bash tools/refresh_profiles_config.sh $c r06 > $OUT/refresh_$c.log 2>&1; done
for h in 3 4; do python bench.py --hidden-res $h --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_o96_hres$h.json 2> /dev/null; done
for h in 3 4; do ANEMOI_CLUSTER_CHAIN=0 python bench.py --hidden-res $h --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $OUT/bench_o96_hres${h}_cluster_off.json 2> /dev/null; done
python bench.py --channels 1024 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_o96_c1024.json 2> /dev/null
for w in 2 4 8; do python tools/rank_floor.py --world $w --hidden-res 5 --wire ipc 2>/dev/null | tail -1 > $OUT/rank_floor_w${w}_r5.json; done
python tools/rank_floor.py --world 8 --hidden-res 6 --wire ipc 2>/dev/null | tail -1 > $OUT/rank_floor_w8_r6.json
for w in 4 8; do ANEMOI_CLUSTER_CHAIN=0 python tools/rank_floor.py --world $w --hidden-res 5 --wire ipc 2>/dev/null | tail -1 > $OUT/rank_floor_w${w}_r5_cluster_off.json; done
ANEMOI_BENCH_TRANSPORT=ipc timeout 600 python bench.py --gpus 8 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT/bench_8ranks_one_gpu_ipc.json 2> $OUT/bench_8ranks_one_gpu_ipc.err
bash tools/r06_pmc_sq.sh r06 o96 > /dev/null 2>&1
bash tools/r06_pmc_sq.sh r06 o96 hres4 --hidden-res 4 > /dev/null 2>&1
python tools/rowchain_time.py > $OUT/rowchain_time.txt 2>&1
ls $OUT | head -60
