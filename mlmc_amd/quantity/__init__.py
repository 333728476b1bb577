"""Quantity tree + estimators: host-side mirror of mlmc/quantity/ of the reference."""
