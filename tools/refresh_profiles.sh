set -x
R=$PWD
mkdir -p $R/gpurun_out/r01c
python bench.py > $R/gpurun_out/r01c/bench.json 2> $R/gpurun_out/r01c/bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o p1 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/r01c/bench_under_rocprof.json 2>/dev/null
python $R/tools/rocprof_summary.py $(find /tmp/p1 -name "*.db" | head -1) > $R/gpurun_out/r01c/kernel_trace_summary.txt 2>&1
TRAIN_GRAPH=0 rocprofv3 --kernel-trace --stats -d /tmp/p2 -o p2 -- python $R/tools/train_step_time.py > $R/gpurun_out/r01c/train_step.txt 2>/dev/null
python $R/tools/rocprof_summary.py $(find /tmp/p2 -name "*.db" | head -1) > $R/gpurun_out/r01c/train_kernel_trace_summary.txt 2>&1
cd $R; python tools/train_step_time.py > $R/gpurun_out/r01c/train_step_plain.txt 2>&1
tail -2 $R/gpurun_out/r01c/train_step_plain.txt
head -c 600 $R/gpurun_out/r01c/bench.json
