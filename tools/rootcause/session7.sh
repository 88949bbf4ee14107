set -x
mkdir -p gpurun_out/rc7
U=$(python tools/time_phase_b.py graph 2>&1 | tail -1); echo "$U" | tee gpurun_out/rc7/phase_b_unprofiled.txt
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/rc7/pb -o b -- python /root/repo/tools/time_phase_b.py graph > /root/repo/gpurun_out/rc7/phase_b_profiled.txt 2>&1)
tail -1 gpurun_out/rc7/phase_b_profiled.txt
D=$(dirname $(find gpurun_out/rc7/pb -name "*kernel_trace.csv" | head -1)); python tools/make_phase_b_summary.py $D gpurun_out/rc7/r06_phase_b_graph_summary.md "$(grep 'Phase B step' gpurun_out/rc7/phase_b_profiled.txt | tail -1)" "$U"
head -12 gpurun_out/rc7/r06_phase_b_graph_summary.md
rm -rf gpurun_out/rc7/pb
bash tools/profile_session.sh --images-per-gpu 8 > gpurun_out/rc7/profile_n8.log 2>&1; tail -3 gpurun_out/rc7/profile_n8.log | cut -c1-300
