set -x
export TMPDIR=/tmp
./tools/lstm_step_prof > gpurun_out/r3_lstm_prof.txt 2>&1; cat gpurun_out/r3_lstm_prof.txt
for w in recipe_tlm recipe_lstm; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$w -o p -- python3 tools/run_workload.py $w 10 > gpurun_out/prof_$w.log 2>&1
tail -1 gpurun_out/prof_$w.log
f=$(find gpurun_out/prof_$w -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r3_${w}_kernel_stats.csv; head -25 $f | cut -c1-200
done
find gpurun_out/prof_recipe_tlm gpurun_out/prof_recipe_lstm -name "*trace.csv" -delete; find gpurun_out -name "*.db" -delete
