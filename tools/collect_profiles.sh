#!/bin/bash
# Run on the GPU box from the repo root (gpurun -- 'bash tools/collect_profiles.sh'): regenerates every file under
# gpurun_out/p that profiles/ keeps for this round.  rocprofv3 gets the program itself after `--` (python3 ...).
set -u
R=$(pwd); P=$R/gpurun_out/p; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $P/prof_b256 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $P/r1_bench_b256_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $P/prof_ft -- python3 $R/bench.py --finetune --train-encoder --batch 32 --steps 5 --warmup 2 > $P/r1_finetune_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d $P/pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $P/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/pmc_tcc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R
for d in FETCH_SIZE WRITE_SIZE sq tcc; do python tools/pmc_summary.py $P/pmc_$d > $P/summary_$d.csv; rm -rf $P/pmc_$d; done
cp $P/prof_b256/*/*_kernel_stats.csv $P/r1_bench_b256_kernel_stats.csv
cp $P/prof_ft/*/*_kernel_stats.csv $P/r1_finetune_b32_trained_encoder_kernel_stats.csv
rm -rf $P/prof_b256 $P/prof_ft
# the bench line reads its `roofline.traffic` from profiles/r1_pmc_{fetch,write}_size_by_kernel.csv: refresh them first
cp $P/summary_FETCH_SIZE.csv $R/profiles/r1_pmc_fetch_size_by_kernel.csv; cp $P/summary_WRITE_SIZE.csv $R/profiles/r1_pmc_write_size_by_kernel.csv
python bench.py > $P/r1_bench_b256.json 2>$P/err1.log
python bench.py --graph --batch 2048 --steps 20 --warmup 20 --no-cpu-baseline > $P/r1_bench_b2048_graph.json 2>/dev/null
python bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline > $P/r1_bench_b1.json 2>/dev/null
python bench.py --encoder small --no-cpu-baseline > $P/r1_bench_b256_dinov2_small.json 2>/dev/null
python bench.py --streams 2 --no-cpu-baseline > $P/r1_bench_b256_two_streams.json 2>/dev/null
python bench.py --finetune --train-encoder --batch 32 --steps 10 --warmup 3 > $P/r1_finetune_b32_trained_encoder.json 2>/dev/null
python bench.py --finetune --batch 32 --steps 10 --warmup 3 > $P/r1_finetune_b32_frozen_encoder.json 2>/dev/null
python bench.py --finetune --batch 256 --steps 10 --warmup 3 > $P/r1_finetune_b256_frozen_encoder.json 2>/dev/null
python tools/blas_ref_bench.py > $P/r1_vendor_gemm_reference.txt 2>/dev/null
HVLA_VARIANTS=5,9 python tools/gemm_bench.py 256 > $P/r1_gemm_isolated.txt 2>/dev/null
HVLA_DBG_MSHRINK=256 HVLA_VARIANTS=9,27 python tools/gemm_bench.py 256 >> $P/r1_gemm_isolated.txt 2>/dev/null   # whole rounds: one launch per tile vs persistent
python tools/bgemm_bench.py > $P/r1_train_gemm_isolated.txt 2>/dev/null
python tools/determinism_probe.py > $P/r1_determinism.txt 2>/dev/null
{ echo "# default (persistent 256x256 + gemm64 tail rows)"; python tools/gemm_race_screen.py 2>/dev/null
  echo "# HVLA_GEMM=ring (lockstep ring kernel, per-lane-column epilogue, W as first MFMA operand)"; HVLA_GEMM=ring python tools/gemm_race_screen.py 2>/dev/null
  echo "# HVLA_NO_PERSIST=1 HVLA_NO_PEEL=1 (one launch per tile, tail rows in the 256x256 grid)"; HVLA_NO_PERSIST=1 HVLA_NO_PEEL=1 python tools/gemm_race_screen.py 2>/dev/null; } > $P/r1_gemm_race_screen.txt
ls $P
