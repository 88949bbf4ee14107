set -x
mkdir -p gpurun_out/rc22
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 3400 python -m pytest tests -m gpu -x -q > gpurun_out/rc22/pytest_gpu.log 2>&1; tail -4 gpurun_out/rc22/pytest_gpu.log
