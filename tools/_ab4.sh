python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent or adversarial" 2>&1 | tail -1
python tools/gpu_src_conv.py 768 432 6; python tools/gpu_src_conv.py 1920 1080 6; python tools/gpu_src_conv.py 3840 2160 5; python tools/gpu_src_conv.py 1280 720 5
python tools/gpu_src_1step.py 1920 1080 256
