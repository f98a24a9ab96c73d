#!/bin/bash
# Run on the GPU box from the repo root (gpurun -- 'bash tools/collect_profiles.sh [round]'): regenerates every file under
# gpurun_out/p that profiles/ keeps for this round.  rocprofv3 gets the program itself after `--` (python3 ...); counter
# passes (--pmc) are separate runs with --kernel-trace only.
set -u
RND=${1:-r6}
R=$(pwd); P=$R/gpurun_out/p; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $P/prof_b256 -- python3 $R/bench.py --steps 10 --warmup 2 --latency-samples 10 --no-cpu-baseline > $P/${RND}_bench_b256_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $P/prof_b1 -- python3 $R/bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $P/prof_ft -- python3 $R/bench.py --finetune --train-encoder --batch 32 --steps 5 --warmup 2 > $P/${RND}_finetune_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d $P/pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --latency-samples 3 --no-cpu-baseline > /dev/null 2>&1; done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $P/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --latency-samples 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/pmc_tcc -- python3 $R/bench.py --steps 3 --warmup 1 --latency-samples 3 --no-cpu-baseline > /dev/null 2>&1
cd $R
for d in FETCH_SIZE WRITE_SIZE sq tcc; do python tools/pmc_summary.py $P/pmc_$d > $P/summary_$d.csv; rm -rf $P/pmc_$d; done
cp $P/prof_b256/*/*_kernel_stats.csv $P/${RND}_bench_b256_kernel_stats.csv
cp $P/prof_ft/*/*_kernel_stats.csv $P/${RND}_finetune_b32_trained_encoder_kernel_stats.csv
cp $P/prof_b1/*/*_kernel_stats.csv $P/${RND}_bench_b1_kernel_stats.csv
rm -rf $P/prof_b256 $P/prof_ft $P/prof_b1
mv $P/summary_FETCH_SIZE.csv $P/${RND}_pmc_fetch_size_by_kernel.csv; mv $P/summary_WRITE_SIZE.csv $P/${RND}_pmc_write_size_by_kernel.csv
mv $P/summary_sq.csv $P/${RND}_pmc_sq_by_kernel.csv; mv $P/summary_tcc.csv $P/${RND}_pmc_tcc_by_kernel.csv
# the bench line reads `roofline.traffic` / `mfma_busy` / `hbm_tbps` of its dominant kernel from profiles/<round>_pmc_*_by_kernel.csv: refresh them first
cp $P/${RND}_pmc_fetch_size_by_kernel.csv $P/${RND}_pmc_write_size_by_kernel.csv $P/${RND}_pmc_sq_by_kernel.csv $P/${RND}_pmc_tcc_by_kernel.csv $R/profiles/
timeout 600 python bench.py > $P/${RND}_bench_b256.json 2>$P/err1.log
timeout 600 python bench.py --graph --batch 2048 --steps 20 --warmup 20 --no-cpu-baseline > $P/${RND}_bench_b2048_graph.json 2>/dev/null
timeout 600 python bench.py --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline > $P/${RND}_bench_b1024.json 2>/dev/null
timeout 600 python bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline > $P/${RND}_bench_b1.json 2>/dev/null
timeout 600 python bench.py --batch 1 --graph --steps 200 --warmup 20 --no-cpu-baseline > $P/${RND}_bench_b1_graph.json 2>/dev/null
timeout 600 python bench.py --encoder small --no-cpu-baseline > $P/${RND}_bench_b256_dinov2_small.json 2>/dev/null
timeout 600 python bench.py --enc-dtype bf16 --no-cpu-baseline > $P/${RND}_bench_b256_bf16.json 2>/dev/null
timeout 600 python bench.py --streams 2 --no-cpu-baseline > $P/${RND}_bench_b256_two_streams.json 2>/dev/null
timeout 600 python bench.py --finetune --train-encoder --batch 32 --steps 10 --warmup 3 > $P/${RND}_finetune_b32_trained_encoder.json 2>/dev/null
timeout 600 python bench.py --finetune --batch 32 --steps 10 --warmup 3 > $P/${RND}_finetune_b32_frozen_encoder.json 2>/dev/null
timeout 600 python bench.py --finetune --batch 256 --steps 10 --warmup 3 > $P/${RND}_finetune_b256_frozen_encoder.json 2>/dev/null
timeout 600 python tools/blas_ref_bench.py > $P/${RND}_vendor_gemm_reference.txt 2>/dev/null
HVLA_VARIANTS=3,2 python tools/gemm_bench.py 256 > $P/${RND}_gemm_isolated.txt 2>/dev/null
timeout 600 python tools/bgemm_bench.py > $P/${RND}_train_gemm_isolated.txt 2>/dev/null
timeout 600 python tools/ctx_phase_times.py 256 > $P/${RND}_ctx_encoder_phases.txt 2>/dev/null
timeout 600 python tools/determinism_probe.py > $P/${RND}_determinism.txt 2>/dev/null
timeout 600 python tools/gemm_race_screen.py > $P/${RND}_gemm_race_screen.txt 2>/dev/null
timeout 300 python tools/lnx_stats.py 256 > $P/${RND}_lnx_stats.txt 2>/dev/null
timeout 300 python tools/lnx_check.py 800 8 9 40 64 85 86 255 256 512 > $P/${RND}_lnx_same_bytes.txt 2>/dev/null
timeout 300 python tools/lnx_check.py 0 8 64 256 512 >> $P/${RND}_lnx_same_bytes.txt 2>/dev/null
timeout 300 python tools/attention_timeline.py 256 > $P/${RND}_attention_timeline.txt 2>/dev/null
timeout 300 python tools/second_order_check.py 16 > $P/${RND}_second_order_inputs.txt 2>/dev/null
timeout 300 python tools/second_order_check.py 16 bf16 >> $P/${RND}_second_order_inputs.txt 2>/dev/null
{ timeout 300 python tools/attention_omean_check.py 16 2>/dev/null | tail -5
  hipcc --offload-arch=gfx950 -O3 tools/permlane_swap_probe.hip -o /tmp/psp 2>/dev/null && timeout 60 /tmp/psp; } > $P/${RND}_attention_mean_rows_and_swap_probe.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "sixty_four or full_geometry_against_golden" 2>/dev/null | grep -E "npz|passed|failed" > $P/${RND}_accuracy.txt
{ hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/policy_hazard_probe.hip -o /tmp/php 2>/dev/null && timeout 300 /tmp/php 1500
  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_war_probe.hip -o /tmp/mwp 2>/dev/null && timeout 120 /tmp/mwp
  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/vmcnt_order_probe.hip -o /tmp/vop 2>/dev/null && timeout 120 /tmp/vop; } > $P/${RND}_policy_hazard_probes.txt 2>&1
ls $P
