"""`from data import face` (reference pyramid.py:6).  On the inference path only the box-coding variances are read
(layers/functions/detection.py:31 -> decode()); the anchor geometry lives in PriorBoxLayer, not in this table."""
face = dict(name='v2', min_dim=640, variance=[0.1, 0.2], clip=False)
