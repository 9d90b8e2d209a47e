#!/bin/bash
# Runs on the GPU box (via gpurun): the default bench line, the rocprofv3 kernel trace of the same command, and the two
# PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, kernel-trace only) the roofline `traffic` figure is derived from.
# Outputs land in gpurun_out/prof_round/; tools/make_profiles.py turns them into the committed summaries under profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# which kernels these counters belong to (bench.py reports `traffic` only for a tree whose csrc hashes to the same value)
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.csrc_sha())" > $O/csrc_sha.txt
# the PMC passes FIRST, their summary written into profiles/ on this box, THEN the bench line: its roofline.traffic then names counters that were
# taken on exactly these kernels in this very run (round 3's set had a bench line from before its PMC passes: traffic null).
# usage: tools/profile_round.sh <tag> [git head]   (the GPU box has no .git: the caller passes the revision)
TAG=${1:-r6_final}
export DVLP_GIT_HEAD=${2:-}
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-kernel-timing --no-cpu-baseline > /dev/null 2> $O/fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 $R/bench.py --steps 2 --warmup 1 --no-kernel-timing --no-cpu-baseline > /dev/null 2> $O/write.err
(cd $R && DVLP_TRAFFIC_ONLY=1 python3 tools/make_profiles.py $TAG > $O/traffic_summary.txt 2>&1)
timeout 900 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python3 $R/bench.py --no-cpu-baseline > $O/trace_bench.json 2> $O/trace.err
# MFMA-busy / LDS-bank-conflict / wait counters of the same command (third PMC pass; SQ + GRBM slots only)
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-kernel-timing --no-cpu-baseline --no-object-tower > /dev/null 2> $O/mfma.err
# K1 alone and the local-loss kernels alone
timeout 300 python3 $R/tools/select_bench.py > $O/select_bench.txt 2>&1
timeout 300 python3 $R/tools/xfused_check.py > $O/xfused_check.txt 2>&1
# the local loss forward / backward alone, the epilogue breakdown of the 256-row kernel, and the stock library on the hot shapes (yardstick)
timeout 300 python3 $R/tools/xloss_bench.py > $O/xloss_bench.txt 2>&1
timeout 300 python3 $R/tools/epi_bench.py > $O/epi_bench.txt 2>&1
timeout 300 python3 $R/tools/lib_gemm_ref.py > $O/lib_gemm_ref.txt 2>&1
DVLP_PROF_REPORT=1 timeout 300 python3 $R/bench.py --no-cpu-baseline --no-object-tower --steps 10 --warmup 3 2>&1 | grep "kern=" | sort > $O/gemm_shapes.txt
# config 5 (32-frame long-video variant, B = 16 per GPU): one bench line and K1 at F = 32; the input side (host staging + PCIe + select)
timeout 600 python3 $R/bench.py --frames 32 --batch 16 --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_f32_b16.json 2> $O/bench_f32.err
# config 4 (configs/ft/msrvtt_o2t-select.json geometry: F = 8, R = 30, per-GPU batch 32 -> 7712 tokens): its own throughput line
timeout 600 python3 $R/bench.py --batch 32 --regions 30 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_f8_r30_b32.json 2> $O/bench_c4.err
# TIMING-ONLY: what the step would take if every forward LayerNorm pass were free (an upper bound for fusing LayerNorm into a neighbouring product)
DVLP_ABLATE_LN_FWD=1 timeout 600 python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/ln_fusion_bound.json 2> $O/ln_bound.err
# (tile height, K split) sweep of the 256-column kernel on the text tower's, config 4's and the object tower's shapes; the local loss' contractions
(timeout 400 python3 $R/tools/tile_sweep.py text; timeout 300 python3 $R/tools/tile_sweep.py c4; timeout 400 python3 $R/tools/tile_sweep.py obj) > $O/tile_sweep.txt 2>&1
timeout 300 python3 $R/tools/loss_gemm_bench.py > $O/loss_gemm_bench.txt 2>&1
SELECT_F=32 SELECT_B=16 timeout 300 python3 $R/tools/select_bench.py > $O/select_bench_f32.txt 2>&1
timeout 300 python3 $R/tools/input_bench.py > $O/input_bench.txt 2>&1
timeout 900 python3 $R/tools/eval_bench.py --pairs 1000 > $O/eval_bench.txt 2>&1
timeout 300 python3 $R/tools/tile_height_bench.py > $O/tile_height_bench.txt 2>&1
timeout 300 python3 $R/tools/attn_bench.py > $O/attn_bench.txt 2>&1
timeout 300 python3 $R/tools/loss_head_bench.py > $O/loss_head_bench.txt 2>&1
# where one replayed step's time goes between the kernels (device-idle gaps, launches under 10 us) -- from the kernel trace above
python3 $R/tools/step_timeline.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 2 > $O/step_timeline.txt 2>&1
# keep what travels back small: the per-dispatch traces are large
for d in trace fetch write mfma; do find $O/$d -name "*kernel_trace.csv" -size +20M -delete; done
ls -la $O $O/trace $O/fetch $O/write $O/mfma
