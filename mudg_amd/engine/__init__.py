"""Executors that run boundary module trees (lvdm.*) on the gfx950 kernels."""
