"""Only `variance` is read on the inference path (reference data/config.py:4-22,
layers/functions/detection.py:31)."""
face = {
    'feature_maps': [160, 80, 40, 20, 10, 5],
    'min_dim': 640,
    'steps': [4, 8, 16, 32, 64, 128],
    'min_sizes': [16, 32, 64, 128, 256, 512],
    'variance': [0.1, 0.2],
    'clip': False,
    'name': 'v2',
}
