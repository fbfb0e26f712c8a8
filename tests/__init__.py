"""Parity and host-logic tests of nerf_sr_amd (pytest; GPU tests carry the `gpu` marker, see conftest.py)."""
