"""Shapes shared by the small-problem timing tools: BASELINE configs[0] and the reference's own datasets
(examples/reproduce_magron2022.py:49-73), plus mid-size squares."""
CASES = [("configs[0] 100x500 K=6", 100, 500, 6), ("animals 50x85 K=4", 50, 85, 4), ("paleo 253x902 K=8", 253, 902, 8),
         ("lastfm 1226x285 K=8", 1226, 285, 8), ("lastfm 1226x285 K=16", 1226, 285, 16), ("1024x1024 K=16", 1024, 1024, 16),
         ("1024x1024 K=32", 1024, 1024, 32), ("2000x2000 K=16", 2000, 2000, 16)]
