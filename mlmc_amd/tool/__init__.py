"""PDF reconstruction tools (mirror of mlmc/tool/simple_distribution.py and mlmc/tool/distribution.py)."""
