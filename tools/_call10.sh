set -u
timeout 900 python -m pytest tests/test_iwad_shapes.py tests/test_golden.py tests/test_gl_readback.py tests/test_gpu_raster_parity.py -x -q -m gpu 2>&1 | tail -3
bash tools/profile_round.sh r04p 2>&1 | tail -25
