set -x
mkdir -p gpurun_out
nproc; cat /sys/fs/cgroup/cpu.max; python -c "import os;print(len(os.sched_getaffinity(0)))"; lscpu | head -20
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 1 --warmup 1 > gpurun_out/bench_r1_b.json 2> gpurun_out/bench_r1_b.err; tail -3 gpurun_out/bench_r1_b.err; cat gpurun_out/bench_r1_b.json
