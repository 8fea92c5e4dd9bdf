cd /tmp
rm -rf /root/repo/gpurun_out/bm_x
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/bm_x -o loops -- python3 /root/repo/tools/loops_anatomy.py c3 c4 > /root/repo/gpurun_out/bm_x.log 2>&1
grep -E "^(c3|c4):" /root/repo/gpurun_out/bm_x.log
python3 /root/repo/tools/rocpd_timeline.py $(find /root/repo/gpurun_out/bm_x -name "*_results.db" | head -n 1) 4 | grep -E "k_build_matrices|k_build_angles"
