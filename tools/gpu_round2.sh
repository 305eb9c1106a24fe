#!/bin/bash
export FMX_ALLOW_KNOBS=1   # the A/B knobs below are development switches: the library ignores them without this
# Round-2 GPU-box visits.  usage (from repo root, through gpurun): bash tools/gpu_round2.sh <tag> <step> [<step> ...]
TAG=${1:-r4}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
pmc_pass() {  # $1 = pass name, $2.. = counters ; workload: tools/pmc_kernels.py
  local name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $O/pmc_$name -o pmc -- python $R/tools/pmc_kernels.py 3 > $O/pmc_$name.log 2>&1)
  tail -2 $O/pmc_$name.log
  find $O/pmc_$name -name '*kernel_trace.csv' -delete
  find $O/pmc_$name -name '*counter_collection.csv' -size +20M -delete
}
for w in "$@"; do
  case $w in
    pmc_4w) for PASS in "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "waves SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
                        "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES" "l2 TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
                        "lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "fetch FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
              set -- $PASS; name=$1; shift
              (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $O/pmc4w_$name -o pmc -- python $R/tools/pmc_gemm4w.py 3 > $O/pmc4w_$name.log 2>&1); tail -1 $O/pmc4w_$name.log
              find $O/pmc4w_$name -name '*kernel_trace.csv' -delete; done
            python tools/pmc_gemm4w_summary.py $O > $O/pmc_gemm4w_summary.json; cat $O/pmc_gemm4w_summary.json;;
    retest) timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q --tb=short -k "knob or batch8" -s 2>&1 | grep -v "^\[parity\] SDXL 1024x1024 batch 8 distinct" | tail -15 > $O/retest.log; cat $O/retest.log;;
    ubench4) tools/ubench/dot2_denorm > $O/dot2_denorm.json 2>&1; cat $O/dot2_denorm.json; tools/ubench/lds_b128_patterns > $O/lds_b128_patterns.jsonl 2>&1; cat $O/lds_b128_patterns.jsonl;;
    ab_attnlib) for L in "" $ABLIB "" $ABLIB; do echo "{\"lib\": \"$L\"}" >> $O/attn_ab.jsonl; FMX_LIB=$L timeout 300 python tools/bench_kernels.py attn 2>> $O/attn_ab.err | grep -v '"force32": true' >> $O/attn_ab.jsonl; done; cat $O/attn_ab.jsonl; tail -2 $O/attn_ab.err;;
    fluxdepth) FMX_PARITY_LOG=$O/parity_flux.jsonl timeout 900 python -m pytest tests/test_gpu_flux.py -m gpu -q -s --tb=short -k "depth or width" 2>&1 | tail -12 > $O/fluxdepth.log; cat $O/fluxdepth.log;;
    attnshort4) timeout 300 python tools/bench_kernels.py attnshort > $O/attnshort.jsonl 2> $O/attnshort.err; cat $O/attnshort.jsonl; tail -2 $O/attnshort.err;;
    solo4w) for E in FMX_GEMM_4W_SOLO=0 FMX_GEMM_4W_SOLO=1; do echo "{\"env\": \"$E\"}" >> $O/solo4w.jsonl; env $E timeout 600 python tools/bench_kernels.py gemm4w 2>> $O/solo4w.err | grep '"tile": 10' >> $O/solo4w.jsonl; done; cat $O/solo4w.jsonl; tail -2 $O/solo4w.err;;
    cputhreads) for T in 16 32 64 128; do FMX_BENCH_CPU_THREADS=$T timeout 600 python bench.py --steps 2 --warmup 1 --no-vae --no-roofline --no-rccl-selfcheck 2>> $O/cputhreads.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'threads': $T, 'cpu_baseline': d['cpu_baseline']}))" >> $O/cputhreads.jsonl; done; cat $O/cputhreads.jsonl;;
    batch64) timeout 900 python bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-rccl-selfcheck > $O/bench_b64.json 2> $O/bench_b64.err; cat $O/bench_b64.json | cut -c1-900; tail -3 $O/bench_b64.err;;
    ab_fa2) for E in FMX_CONV_FASTADDR=3 FMX_CONV_FASTADDR=1 FMX_CONV_FASTADDR=3 FMX_CONV_FASTADDR=1; do env $E timeout 600 python bench.py --no-cpu-baseline --no-rccl-selfcheck --steps 10 --breakdown $O/bd_$E.jsonl --vae-breakdown $O/vbd_$E.jsonl 2>> $O/abfa2.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'env':'$E','ms_per_step':d['ms_per_step'],'vae_ms':d['vae_decode_ms_per_batch'],'gemm_tflops':d['roofline']['achieved'],'sclk':(d.get('clocks_during_timed_steps') or {}).get('sclk_mhz',{}).get('mean')}))" >> $O/abfa2.jsonl; done; cat $O/abfa2.jsonl; tail -2 $O/abfa2.err; grep -h " up" $O/bd_*.jsonl $O/vbd_*.jsonl | cut -c1-200;;
    gemm4w) timeout 900 python tools/bench_kernels.py gemm4w > $O/gemm4w.jsonl 2> $O/gemm4w.err; cat $O/gemm4w.jsonl; tail -3 $O/gemm4w.err;;
    ktests4w) timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rccl.py tests/test_gpu_vae_bf16.py -m gpu -q --tb=short -k "two_workgroup or linear_plain or gemm256 or knob or rccl or overflow or layernorm_folded" 2>&1 | tail -40 > $O/ktests4w.log; tail -25 $O/ktests4w.log;;
    ab_4w) for E in FMX_GEMM_4W=0 FMX_GEMM_4W=2 FMX_GEMM_4W=0 FMX_GEMM_4W=2; do env $E timeout 600 python bench.py --no-cpu-baseline --no-vae --no-rccl-selfcheck --steps 10 --breakdown $O/breakdown_$E.jsonl 2>> $O/ab4w.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'env':'$E','ms_per_step':d['ms_per_step'],'gemm_tflops':d['roofline']['achieved'],'gemm_ms':d['roofline']['kernel_time_per_forward_ms'],'attn':d['roofline_attention']['achieved'],'sclk':(d.get('clocks_during_timed_steps') or {}).get('sclk_mhz'),'power':(d.get('clocks_during_timed_steps') or {}).get('power_w'),'knobs':d.get('knobs')}))" >> $O/ab4w.jsonl; done; cat $O/ab4w.jsonl; tail -3 $O/ab4w.err;;
    tests) FMX_PARITY_LOG=$O/parity.jsonl timeout 1500 python -m pytest tests -m gpu -q -s --tb=line 2>&1 | grep -v "^\[parity\]" | tail -120 > $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log;;
    attnshort) for E in "FMX_ATTN_SHORT=0" "FMX_ATTN_SHORT=1" "FMX_ATTN_SHORT=2"; do env $E timeout 300 python tools/bench_kernels.py attnshort >> $O/attnshort.jsonl 2>> $O/attnshort.err; done; cat $O/attnshort.jsonl; tail -3 $O/attnshort.err;;
    atests) timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "attention" 2>&1 | tail -40 > $O/atests.log; tail -15 $O/atests.log;;
    atests2) FMX_ATTN_SHORT=2 timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "attention" 2>&1 | tail -40 > $O/atests2.log; tail -15 $O/atests2.log;;
    ab_persist) for E in "FMX_GEMM_PERSIST=0" "FMX_GEMM_PERSIST=1" "FMX_GEMM_PERSIST=0" "FMX_GEMM_PERSIST=1"; do env $E timeout 600 python bench.py --no-cpu-baseline --no-vae --steps 10 2>> $O/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'env':'$E','ms_per_step':d['ms_per_step'],'gemm_tflops':d['roofline']['achieved'],'gemm_ms':d['roofline']['kernel_time_per_forward_ms'],'attn':d['roofline_attention']['achieved'],'xattn_ms':d['roofline_attention_short_keys']['kernel_time_per_forward_ms'],'clocks':d.get('clocks_during_timed_steps')}))" >> $O/ab.jsonl; done; cat $O/ab.jsonl; tail -3 $O/ab.err;;
    clock) for R in 0 1; do FMX_LIB=tools/_build/libfmx_clock.so FMX_TILE=7 FMX_CLOCK_RESIDUAL=$R timeout 300 python tools/clock_gemm.py >> $O/clock.txt 2>> $O/clock.err; done; cat $O/clock.txt; tail -2 $O/clock.err;;
    clock16) for MFV in 16 32 16 32; do echo "== FMX_GEMM_MFMA=$MFV" >> $O/clock16.txt; FMX_GEMM_MFMA=$MFV FMX_LIB=tools/_build/libfmx_clock.so FMX_TILE=7 timeout 300 python tools/clock_gemm.py >> $O/clock16.txt 2>> $O/clock16.err; done; cat $O/clock16.txt; tail -2 $O/clock16.err;;
    ab_lib) for L in "" $ABLIB "" $ABLIB; do FMX_LIB=$L timeout 600 python bench.py --no-cpu-baseline --no-vae --steps 10 2>> $O/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'lib':'$L','ms_per_step':d['ms_per_step'],'gemm_tflops':d['roofline']['achieved'],'gemm_ms':d['roofline']['kernel_time_per_forward_ms'],'attn':d['roofline_attention']['achieved'],'xattn_ms':d['roofline_attention_short_keys']['kernel_time_per_forward_ms'],'sclk':(d.get('clocks_during_timed_steps') or {}).get('sclk_mhz')}))" >> $O/ab.jsonl; done; cat $O/ab.jsonl; tail -3 $O/ab.err;;
    timeline) for X in 1 0; do FMX_GEMM_XTILE=$X FMX_LIB=tools/_build/libfmx_timeline.so timeout 300 python tools/tile_timeline.py >> $O/timeline.txt 2>> $O/timeline.err; done; cat $O/timeline.txt | cut -c1-250; tail -3 $O/timeline.err;;
    ab_env) for E in $ABENV0 $ABENV1 $ABENV0 $ABENV1; do env ${E//,/ } timeout 600 python bench.py --no-cpu-baseline --no-vae --steps 10 2>> $O/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'env':'$E','ms_per_step':d['ms_per_step'],'gemm_tflops':d['roofline']['achieved'],'gemm_ms':d['roofline']['kernel_time_per_forward_ms'],'attn':d['roofline_attention']['achieved'],'xattn_ms':d['roofline_attention_short_keys']['kernel_time_per_forward_ms'],'sclk':(d.get('clocks_during_timed_steps') or {}).get('sclk_mhz')}))" >> $O/ab.jsonl; done; cat $O/ab.jsonl; tail -3 $O/ab.err;;
    e2etests) timeout 1200 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_controlnet.py tests/test_gpu_hooks.py -m gpu -q --tb=short 2>&1 | tail -30 > $O/e2e.log; tail -12 $O/e2e.log;;
    gnapply) for E in "FMX_GN_BLOCK_KB=64" "FMX_GN_BLOCK_KB=16" "FMX_GN_BLOCK_KB=8" "FMX_GN_BLOCK_KB=12" "FMX_GN_BLOCK_KB=16 FMX_GN_VARIANT=2" "FMX_GN_BLOCK_KB=8 FMX_GN_VARIANT=2" "FMX_GN_BLOCK_KB=16"; do env $E timeout 300 python tools/bench_kernels.py gnapply >> $O/gnapply.jsonl 2>> $O/gnapply.err; done; cat $O/gnapply.jsonl; tail -3 $O/gnapply.err;;
    soak) timeout 600 python tools/soak.py 3 8 > $O/soak.jsonl 2> $O/soak.err; tail -8 $O/soak.jsonl; tail -2 $O/soak.err;;
    ab_sd15) for E in "FMX_X=0" "FMX_GEMM_PERSIST=0" "FMX_ATTN_SHORT=0" "FMX_GN_BLOCK_KB=64" "FMX_X=0" "FMX_GEMM_PERSIST=0 FMX_ATTN_SHORT=0 FMX_GN_BLOCK_KB=64"; do env $E timeout 600 python bench.py --config ${ABCFG:-sd15-b4-eulera} --no-cpu-baseline --no-vae --steps 20 2>> $O/ab_sd15.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'env':'$E','cfg':d['config']['name'],'ms_per_step':d['ms_per_step'],'gemm_ms':d['roofline']['kernel_time_per_forward_ms'],'attn_ms':d['roofline_attention']['kernel_time_per_forward_ms'],'xattn_ms':(d.get('roofline_attention_short_keys') or {}).get('kernel_time_per_forward_ms'),'gn_ms':d['roofline_groupnorm']['kernel_time_per_forward_ms'],'sclk':(d.get('clocks_during_timed_steps') or {}).get('sclk_mhz',{}).get('mean')}))" >> $O/ab_sd15.jsonl; done; cat $O/ab_sd15.jsonl; tail -3 $O/ab_sd15.err;;
    attn512) for E in FMX_ATTN512_SLICES=4 FMX_ATTN512_SLICES=2 FMX_ATTN512_SLICES=4 FMX_ATTN512_SLICES=2; do env $E timeout 300 python tools/bench_kernels.py attn512 >> $O/attn512.jsonl 2>> $O/attn512.err; done; cat $O/attn512.jsonl; tail -2 $O/attn512.err;;
    vtests) timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vae_bf16.py tests/test_gpu_boundary.py -m gpu -q --tb=short -k "512 or vae or single_head or spatial or decode" 2>&1 | tail -15 > $O/vtests.log; tail -8 $O/vtests.log;;
    ktests16) FMX_GEMM_MFMA=16 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_boundary.py -m gpu -q --tb=short 2>&1 | tail -40 > $O/ktests16.log; tail -15 $O/ktests16.log;;
    epi16) for E in FMX_GEMM_MFMA=16 FMX_GEMM_MFMA=32 FMX_GEMM_MFMA=16 FMX_GEMM_MFMA=32; do env $E timeout 300 python tools/bench_kernels.py epi >> $O/epi16.jsonl 2>> $O/epi16.err; done; tail -2 $O/epi16.err;;
    ab_vae) for E in FMX_GEMM_MFMA=32 FMX_GEMM_MFMA=16 FMX_GEMM_MFMA=32 FMX_GEMM_MFMA=16; do env $E timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 4 --warmup 1 2>> $O/ab_vae.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'env':'$E','ms_per_step':d['ms_per_step'],'vae_ms':d['vae_decode_ms_per_batch']}))" >> $O/ab_vae.jsonl; done; cat $O/ab_vae.jsonl; tail -2 $O/ab_vae.err;;
    clocksweep) for L in $SWEEPLIBS; do echo "== $L" >> $O/clocksweep.txt; FMX_LIB=tools/_build/libfmx_$L.so FMX_TILE=7 timeout 300 python tools/clock_gemm.py 2>> $O/clocksweep.err | grep "TF/s" >> $O/clocksweep.txt; done; cat $O/clocksweep.txt; tail -2 $O/clocksweep.err;;
    dual) timeout 600 python tools/bench_kernels.py dual > $O/dual.jsonl 2> $O/dual.err; cat $O/dual.jsonl; tail -3 $O/dual.err;;
    epi) for L in $EPILIBS; do FMX_LIB=$L timeout 300 python tools/bench_kernels.py epi >> $O/epi.jsonl 2>> $O/epi.err; done; cat $O/epi.jsonl;;
    testsx) FMX_PARITY_LOG=$O/parity.jsonl timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log;;
    ktests) timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -40 > $O/ktests.log; tail -12 $O/ktests.log;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log;;
    bench) timeout 1200 python bench.py --breakdown $O/breakdown.jsonl --vae-breakdown $O/vae_breakdown.jsonl > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err;;
    bench_nofuse) FMX_GN_FUSED_STATS=0 timeout 900 python bench.py --no-cpu-baseline --breakdown $O/breakdown_nofuse.jsonl > $O/bench_nofuse.json 2> $O/bench_nofuse.err; cat $O/bench_nofuse.json; tail -3 $O/bench_nofuse.err;;
    bench_cfgs) for c in sd15-b4-eulera sdxl-b8-dpmpp2m30-vae flux-b2-bf16; do timeout 900 python bench.py --config $c --steps 8 > $O/bench_$c.json 2> $O/bench_$c.err; cat $O/bench_$c.json; tail -2 $O/bench_$c.err; done;;
    gnbench) timeout 600 python tools/bench_kernels.py gn > $O/gnbench.log 2>&1; cat $O/gnbench.log | tail -30;;
    attnbench) timeout 600 python tools/bench_kernels.py attn > $O/attnbench.log 2>&1; tail -20 $O/attnbench.log;;
    prof) (cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o kt -- \
            python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_bench.log 2>&1); tail -2 $O/prof_bench.log;
          find $O/prof -name '*kernel_trace.csv' -size +20M -delete; ls -la $O/prof/* | head;;
    counters) (cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 -L > $O/counters_all.txt 2>&1); grep -i -o -E "\b(SQ_[A-Z0-9_]*(MFMA|BUSY|WAVE_CYCLES|INSTS_VALU|ACTIVE_INST|WAIT)[A-Z0-9_]*|GRBM_GUI_ACTIVE|GRBM_COUNT)\b" $O/counters_all.txt | sort -u > $O/counters_sq.txt; wc -l $O/counters_sq.txt; head -80 $O/counters_sq.txt;;
    pmc_mfma) pmc_pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE;
              pmc_pass waves SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA;;
    pmc_hbm) for C in FETCH_SIZE WRITE_SIZE; do (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C -f csv -d $O/pmc_$C -o pmc -- \
            python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-roofline --no-graph > $O/pmc_$C.log 2>&1); tail -1 $O/pmc_$C.log; done;
          python $R/tools/pmc_summary.py $O > $O/pmc_summary.json; cat $O/pmc_summary.json; find $O -name '*counter_collection.csv' -size +30M -delete; find $O -name '*kernel_trace.csv' -delete;;
  esac
done
