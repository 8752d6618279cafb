timeout 600 python -m pytest tests/test_iwad_shapes.py -x -q -m gpu 2>&1 | tail -40
