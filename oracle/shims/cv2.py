"""ORACLE SCAFFOLDING (tests only) -- empty stand-in so `utils/visualize_utils.py:2` imports."""
