"""Helpers of bench.py (measurement only; nothing here is product code): the extra objects of the bench line live in modules of their own."""
