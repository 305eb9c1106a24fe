#!/bin/bash
export FMX_ALLOW_KNOBS=1   # the A/B knobs below are development switches: the library ignores them without this
# Round-6 GPU-box visits.  usage (from repo root, through gpurun): bash tools/gpu_round6.sh <tag> <step> [<step> ...]
TAG=${1:-r40}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
benchline() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(json.dumps({'tag':'$1','ms_per_step':d['ms_per_step'],'it_s':d['value'],'vae_ms':d.get('vae_decode_ms_per_batch'),'gemm_tflops':(d.get('roofline') or {}).get('achieved'),'gemm_ms':(d.get('roofline') or {}).get('kernel_time_per_forward_ms'),'attn':(d.get('roofline_attention') or {}).get('achieved'),'attn_ms':(d.get('roofline_attention') or {}).get('kernel_time_per_forward_ms'),'gn_ms':(d.get('roofline_groupnorm') or {}).get('kernel_time_per_forward_ms'),'sclk':((d.get('clocks_during_timed_steps') or {}).get('sclk_mhz') or {}).get('mean'),'knobs':d.get('knobs')}))"; }
for w in "$@"; do
  case $w in
    newtests) FMX_PARITY_LOG=$O/parity_new.jsonl timeout 1500 python -m pytest tests/test_gpu_sharp_parity.py tests/test_checkpoint_file.py tests/test_gpu_lora.py -m gpu -q -s --tb=short --durations=12 -k "bench_batch or inside_the_layernorm or checkpoint or bfloat16_storage or flux_lora" 2>&1 | grep -v "^\[parity\]" | tail -60 > $O/newtests.log; tail -30 $O/newtests.log | cut -c1-1500;;
    fluxjob) FMX_PARITY_LOG=$O/parity_fluxjob.jsonl timeout 900 python -m pytest tests/test_gpu_flux.py -m gpu -q -s --tb=short -k "job_at_full_depth or bf16_flux_forward_and_sampling" 2>&1 | tail -12 > $O/fluxjob.log; cat $O/fluxjob.log | cut -c1-900;;
    sharp) FMX_PARITY_LOG=$O/sharp_parity.jsonl timeout 1500 python -m pytest tests/test_gpu_sharp_parity.py -m gpu -q -s --tb=short --durations=10 2>&1 | grep -v "^\[parity\]" | tail -40 > $O/sharp.log; tail -25 $O/sharp.log | cut -c1-1200;;
    tests) FMX_PARITY_LOG=$O/parity.jsonl timeout 1700 python -m pytest tests -m gpu -q -s --tb=line --durations=25 2>&1 | grep -v "^\[parity\]" | tail -150 > $O/pytest_gpu.log; tail -40 $O/pytest_gpu.log | cut -c1-400;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log;;
    bench) timeout 1500 python bench.py --breakdown $O/breakdown.jsonl --vae-breakdown $O/vae_breakdown.jsonl > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err;;
    torchbase) timeout 600 python bench.py --torch-rocm-baseline > $O/torchbase.log 2>&1; tail -5 $O/torchbase.log | cut -c1-600;;
    benchflux) timeout 900 python bench.py --config flux-b2-bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-rccl-selfcheck --breakdown $O/breakdown_flux.jsonl > $O/bench_flux.json 2> $O/bench_flux.err; cat $O/bench_flux.json | cut -c1-1800; tail -2 $O/bench_flux.err; head -12 $O/breakdown_flux.jsonl;;
    vaebench) timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rccl-selfcheck --no-other-configs --no-torch-rocm-baseline --no-roofline --vae-breakdown $O/vae_breakdown.jsonl 2>> $O/vaebench.err | benchline vae >> $O/vaebench.jsonl; cat $O/vaebench.jsonl; head -30 $O/vae_breakdown.jsonl;;
    ab_env) for E in $ABENV0 $ABENV1 $ABENV0 $ABENV1; do env ${E//,/ } timeout 600 python bench.py ${ABFLAGS:---no-vae} --no-cpu-baseline --no-rccl-selfcheck --no-other-configs --no-torch-rocm-baseline --steps 10 2>> $O/ab.err | benchline "$E" >> $O/ab.jsonl; done; cat $O/ab.jsonl; tail -3 $O/ab.err;;
    ab_lib) for L in "" $ABLIB "" $ABLIB; do FMX_LIB=$L timeout 600 python bench.py ${ABFLAGS:---no-vae} --no-cpu-baseline --no-rccl-selfcheck --no-other-configs --no-torch-rocm-baseline --steps 10 2>> $O/ab.err | benchline "lib=$L" >> $O/ab.jsonl; done; cat $O/ab.jsonl; tail -3 $O/ab.err;;
    kb) timeout 900 python tools/bench_kernels.py $KB > $O/kb_$KB.jsonl 2> $O/kb_$KB.err; cat $O/kb_$KB.jsonl | cut -c1-400; tail -3 $O/kb_$KB.err;;
    attnrot) FMX_ATTN_WS_ROT=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_flux.py tests/test_gpu_flux_sharp_parity.py -m gpu -q --tb=short -k "d128 or attention or flux_forward_at_its_own_width or tiny or sharp or stage" 2>&1 | tail -15 > $O/attnrot_tests.log; tail -6 $O/attnrot_tests.log | cut -c1-300;
             for E in 0 1 0 1; do FMX_ATTN_WS_ROT=$E timeout 300 python tools/bench_kernels.py attn128 >> $O/attnrot.jsonl 2>> $O/attnrot.err; done; cat $O/attnrot.jsonl | cut -c1-300; tail -2 $O/attnrot.err;;
    pmc_attnrot) for E in 0 1; do mkdir -p $O/rot$E; for PASS in "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "waves SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"; do
               set -- $PASS; name=$1; shift
               (cd /tmp && export TMPDIR=/tmp && FMX_ATTN_WS_ROT=$E timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $O/rot$E/pmc_$name -o pmc -- python $R/tools/pmc_attn128.py > $O/rot$E/pmc_$name.log 2>&1); tail -1 $O/rot$E/pmc_$name.log | cut -c1-200
               find $O/rot$E/pmc_$name -name '*kernel_trace.csv' -delete; done
             python tools/pmc_mfma_summary.py $O/rot$E > $O/pmc_attn128_rot$E.json; cat $O/pmc_attn128_rot$E.json | head -40; done;;
    pmc_py) mkdir -p $O/$PMCTAG; for PASS in "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "waves SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES"; do
               set -- $PASS; name=$1; shift
               (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $O/$PMCTAG/pmc_$name -o pmc -- python $R/$PMCPY > $O/$PMCTAG/pmc_$name.log 2>&1); tail -1 $O/$PMCTAG/pmc_$name.log | cut -c1-200
               find $O/$PMCTAG/pmc_$name -name '*kernel_trace.csv' -delete; done
             python tools/pmc_mfma_summary.py $O/$PMCTAG > $O/pmc_$PMCTAG.json; cat $O/pmc_$PMCTAG.json | head -60;;
    prof_flux) (cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_flux -o kt -- \
            python $R/bench.py --config flux-b2-bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-rccl-selfcheck --no-roofline > $O/prof_flux_bench.log 2>&1); tail -1 $O/prof_flux_bench.log | cut -c1-300;
          find $O/prof_flux -name '*kernel_trace.csv' -delete; head -25 $O/prof_flux/kt_kernel_stats.csv | cut -c1-220;;
    kbvar) for L in "" $KBLIBS; do FMX_LIB=$L timeout 600 python tools/bench_kernels.py $KB >> $O/kbvar_$KB.jsonl 2>> $O/kbvar_$KB.err; done; cat $O/kbvar_$KB.jsonl | cut -c1-500; tail -3 $O/kbvar_$KB.err;;
    pyt) eval timeout 1200 python -m pytest $PYT -m gpu -q --tb=short -s 2>&1 | grep -v "^\[parity\]" | tail -40 > $O/pyt.log; tail -30 $O/pyt.log | cut -c1-600;;
    prof) (cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o kt -- \
            python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-torch-rocm-baseline --no-rccl-selfcheck > $O/prof_bench.log 2>&1); tail -2 $O/prof_bench.log | cut -c1-600;
          find $O/prof -name '*kernel_trace.csv' -size +20M -delete; ls -la $O/prof/* | head;;
    pmc_hbm) for C in FETCH_SIZE WRITE_SIZE; do (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C -f csv -d $O/pmc_$C -o pmc -- \
            python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-roofline --no-graph --no-other-configs --no-torch-rocm-baseline --no-rccl-selfcheck > $O/pmc_$C.log 2>&1); tail -1 $O/pmc_$C.log | cut -c1-300; done;
          python $R/tools/pmc_summary.py $O > $O/pmc_summary.json; cat $O/pmc_summary.json | cut -c1-1500; find $O -name '*counter_collection.csv' -size +30M -delete; find $O -name '*kernel_trace.csv' -delete;;
  esac
done
