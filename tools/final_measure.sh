#!/bin/bash
# The round's measurement pass on one GPU box: default bench, rocprofv3 kernel table, the two PMC traffic passes, and the
# other modes / the 512-ray share.  Writes under gpurun_out/final/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "default bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --no-cfg2 > $O/bench_under_rocprof.json 2> $O/kt.err
echo "kernel trace done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-inference --no-cfg2 > $O/pmc_f.json 2> $O/pmc_f.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-inference --no-cfg2 > $O/pmc_w.json 2> $O/pmc_w.err
python3 $R/tools/pmc_summary.py $O/f/f_counter_collection.csv $O/w/w_counter_collection.csv "$(hostname) $(date -u +%Y-%m-%dT%H:%MZ), one MI355X, bench.py --steps 1 --warmup 1 (global batch 4096 rays x 128+128 samples, fused_f16x2)" 2 > $O/pmc_summary.json 2> $O/pmc_summary.err || true
# BASELINE configs[1] leg (plain bf16, 256 x 512 pool, 512-ray batches): the same two passes in that mode at that size
C2="--mlp-mode fused_bf16 --global-batch 512 --height 256 --width 512 --graph off --steps 1 --warmup 1 --no-cpu-baseline --no-inference --no-cfg2"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f2 -o f -- python3 $R/bench.py $C2 > $O/pmc_f2.json 2> $O/pmc_f2.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w2 -o w -- python3 $R/bench.py $C2 > $O/pmc_w2.json 2> $O/pmc_w2.err
python3 $R/tools/pmc_summary.py $O/f2/f_counter_collection.csv $O/w2/w_counter_collection.csv "$(hostname) $(date -u +%Y-%m-%dT%H:%MZ), one MI355X, bench.py $C2" 2 > $O/pmc_summary_cfg2.json 2> $O/pmc_summary_cfg2.err || true
cd $R && tools/pmc_pass.sh sq 2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT
echo "pmc done"
cd $R
timeout -k 10 200 python3 bench.py --global-batch 512 --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-cfg2 > $O/bench_b512.json 2> $O/bench_b512.err
for m in fused fused_f16x2_t32 fused_bf16 layerwise; do
  timeout -k 10 200 python3 bench.py --mlp-mode $m --steps 10 --warmup 3 --no-cpu-baseline --no-cfg2 > $O/bench_$m.json 2> $O/bench_$m.err
done
echo "modes done"
