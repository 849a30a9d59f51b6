# copies gpurun_out/closing4/* into profiles/r04_<tag>_* (run in the build container after bash tools/closing_r04.sh <tag> on the GPU box)
T=${1:-a}; S=gpurun_out/closing4; P=profiles
cp $S/bench_driver_form.json $P/r04_${T}_bench_driver_form.json
cp $S/bench.json $P/r04_${T}_bench.json
cp $S/pipelined_kernel_stats.csv $P/r04_${T}_kernel_stats.csv
cp $S/one_frame_kernel_stats.csv $P/r04_${T}_one_frame_at_a_time_kernel_stats.csv
cp $S/one_frame_kernel_stats.csv $P/in_frame_kernel_stats.csv
cp $S/trace_sequence.txt $P/r04_${T}_trace_sequence.txt
cp $S/trace_overlap.txt $P/r04_${T}_trace_overlap.txt
cp $S/waymo.json $P/r04_${T}_waymo_bench.json
cp $S/waymo_kernel_stats.csv $P/r04_${T}_waymo_kernel_stats.csv
cp $S/waymo_one_frame_kernel_stats.csv $P/r04_${T}_waymo_one_frame_at_a_time_kernel_stats.csv
cp $S/waymo_one_frame_kernel_stats.csv $P/in_frame_kernel_stats_waymo.csv
cp $S/waymo_trace_sequence.txt $P/r04_${T}_waymo_trace_sequence.txt
cp $S/train.json $P/r04_${T}_train_bench.json
cp $S/train_kernel_stats.csv $P/r04_${T}_train_kernel_stats.csv
cp $S/pvrcnn.json $P/r04_${T}_pvrcnn_bench.json
cp $S/pvrcnn_e2e.json $P/r04_${T}_pvrcnn_e2e_bench.json
[ -f $S/pvrcnn_kernel_stats.csv ] && cp $S/pvrcnn_kernel_stats.csv $P/r04_${T}_pvrcnn_stage2_kernel_stats.csv
[ -f $S/pvrcnn_e2e_kernel_stats.csv ] && cp $S/pvrcnn_e2e_kernel_stats.csv $P/r04_${T}_pvrcnn_e2e_kernel_stats.csv
[ -f $S/pmc_traffic.txt ] && cp $S/pmc_traffic.txt $P/r04_pmc_traffic.txt && cp $S/pmc_traffic.json $P/pmc_traffic.json
[ -f $S/pmc_traffic_waymo.txt ] && cp $S/pmc_traffic_waymo.txt $P/r04_pmc_traffic_waymo.txt && cp $S/pmc_traffic_waymo.json $P/pmc_traffic_waymo.json
[ -f $S/mb_sparse_layers.txt ] && cp $S/mb_sparse_layers.txt $P/r04_${T}_mb_sparse_layers.txt
ls $P | grep "r04_${T}_" | wc -l
