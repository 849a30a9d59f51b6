# copies gpurun_out/closing5/* into profiles/r05_<tag>_* (run in the build container after bash tools/closing_r05.sh <tag> on the GPU box)
T=${1:-a}; S=gpurun_out/closing5; P=profiles
cp $S/bench_driver_form.json $P/r05_${T}_bench_driver_form.json
cp $S/bench.json $P/r05_${T}_bench.json
cp $S/pipelined_kernel_stats.csv $P/r05_${T}_kernel_stats.csv
cp $S/one_frame_kernel_stats.csv $P/r05_${T}_one_frame_at_a_time_kernel_stats.csv
cp $S/one_frame_kernel_stats.csv $P/in_frame_kernel_stats.csv
cp $S/trace_sequence.txt $P/r05_${T}_trace_sequence.txt
cp $S/trace_overlap.txt $P/r05_${T}_trace_overlap.txt
cp $S/waymo.json $P/r05_${T}_waymo_bench.json
cp $S/waymo_kernel_stats.csv $P/r05_${T}_waymo_kernel_stats.csv
cp $S/waymo_one_frame_kernel_stats.csv $P/r05_${T}_waymo_one_frame_at_a_time_kernel_stats.csv
cp $S/waymo_one_frame_kernel_stats.csv $P/in_frame_kernel_stats_waymo.csv
cp $S/waymo_trace_sequence.txt $P/r05_${T}_waymo_trace_sequence.txt
cp $S/train.json $P/r05_${T}_train_bench.json
cp $S/train_kernel_stats.csv $P/r05_${T}_train_kernel_stats.csv
cp $S/pvrcnn.json $P/r05_${T}_pvrcnn_bench.json
cp $S/pvrcnn_e2e.json $P/r05_${T}_pvrcnn_e2e_bench.json
[ -f $S/pvrcnn_kernel_stats.csv ] && cp $S/pvrcnn_kernel_stats.csv $P/r05_${T}_pvrcnn_stage2_kernel_stats.csv
[ -f $S/pvrcnn_e2e_kernel_stats.csv ] && cp $S/pvrcnn_e2e_kernel_stats.csv $P/r05_${T}_pvrcnn_e2e_kernel_stats.csv
[ -f $S/pmc_traffic.txt ] && cp $S/pmc_traffic.txt $P/r05_pmc_traffic.txt && cp $S/pmc_traffic.json $P/pmc_traffic.json
[ -f $S/pmc_traffic_waymo.txt ] && cp $S/pmc_traffic_waymo.txt $P/r05_pmc_traffic_waymo.txt && cp $S/pmc_traffic_waymo.json $P/pmc_traffic_waymo.json
[ -f $S/mb_prec_ab.txt ] && cp $S/mb_prec_ab.txt $P/r05_${T}_mb_prec_ab.txt
cp $S/plumbing.json $P/r05_${T}_plumbing_bench.json
cp $S/bench_bf16x3.json $P/r05_${T}_bf16x3_bench.json
ls $P | grep "r05_${T}_" | wc -l
