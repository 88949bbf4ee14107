set -x
mkdir -p gpurun_out/rc5
timeout 900 python -m pytest -m gpu -x -q -s "tests/test_gpu_generator.py::test_graph_full_conditioned_and_heavy_tailed" "tests/test_gpu_ops.py::test_sampler_indices_exact" "tests/test_gpu_loops.py::test_config_c2_at_full_size_trajectory" > gpurun_out/rc5/pytest_s.log 2>&1; grep -h "conditioned\|range\|sampler indices\|C2 full size\|passed\|failed\|Error\|assert" gpurun_out/rc5/pytest_s.log | cut -c1-250 | head -40
EG3D_DETERMINISTIC=1 timeout 900 python -m pytest -m gpu -x -q -s "tests/test_gpu_generator.py::test_graph_full_conditioned_and_heavy_tailed" "tests/test_gpu_ops.py::test_sampler_indices_exact" > gpurun_out/rc5/pytest_s_det.log 2>&1; grep -h "conditioned\|range\|sampler indices\|passed\|failed" gpurun_out/rc5/pytest_s_det.log | cut -c1-250 | head -40
for v in 0 1 0 1; do echo "CONV_WS_TRAINABLE=$v: $(EG3D_CONV_WS_TRAINABLE=$v python tools/time_phase_b.py graph 2>&1 | tail -1)"; done | tee gpurun_out/rc5/phase_b_ab.log
bash tools/pmc_membound.sh > gpurun_out/rc5/pmc.log 2>&1; tail -3 gpurun_out/rc5/pmc.log
bash tools/profile_session.sh > gpurun_out/rc5/profile_session.log 2>&1; tail -5 gpurun_out/rc5/profile_session.log | cut -c1-300
