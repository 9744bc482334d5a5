python -m pytest tests -x -q -m gpu --tb=short 2>&1 | tail -6
python scripts/step_attrib.py 2>&1 | grep -v "amdgpu\|Warn\|warn" > gpurun_out/step_attrib_now.txt
