set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 1 --warmup 1 > gpurun_out/bench_r1_a.json 2> gpurun_out/bench_r1_a.err; tail -3 gpurun_out/bench_r1_a.err; cat gpurun_out/bench_r1_a.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_a -o p -- python $R/bench.py --steps 1 --warmup 0 --ddpm-steps 100 --no-cpu-baseline > $R/gpurun_out/prof_a.log 2>&1
cd $R; ls -la gpurun_out/prof_a | head; tail -3 gpurun_out/prof_a.log
