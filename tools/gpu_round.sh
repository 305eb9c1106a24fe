#!/bin/bash
# One GPU-box visit: parity tests, headline bench, rocprofv3 kernel-trace stats of the same bench command.
# usage (from repo root, through gpurun): bash tools/gpu_round.sh <tag> [tests|bench|prof ...]
TAG=${1:-r1}; shift
WHAT=${@:-tests bench prof}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for w in $WHAT; do
  case $w in
    tests) timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -150 > $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log;;
    bench) timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err;;
    prof) (cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o kt -- \
            python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_bench.log 2>&1); tail -2 $O/prof_bench.log;
          find $O/prof -name '*kernel_trace.csv' -size +20M -delete; ls -la $O/prof/* | head;;
    pmc) for C in FETCH_SIZE WRITE_SIZE; do (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C -f csv -d $O/pmc_$C -o pmc -- \
            python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-roofline --no-graph > $O/pmc_$C.log 2>&1); tail -1 $O/pmc_$C.log; done;
          python $R/tools/pmc_summary.py $O > $O/pmc_summary.json; cat $O/pmc_summary.json; find $O -name '*counter_collection.csv' -size +30M -delete; find $O -name '*kernel_trace.csv' -delete;;
    kbench) timeout 900 python tools/bench_kernels.py > $O/kbench.log 2>&1; tail -60 $O/kbench.log;;
    gemmbench) timeout 900 python tools/bench_kernels.py gemm > $O/gemmbench.log 2>&1; tail -100 $O/gemmbench.log;;
    ktests) timeout -s KILL 240 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm256_identity or gemm256_epilogues" > $O/ktests_quick.log 2>&1 || { tail -30 $O/ktests_quick.log; echo "QUICK TEST FAILED -- stopping"; exit 1; }
            timeout -s KILL 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -60 > $O/ktests.log; tail -25 $O/ktests.log;;
  esac
done
