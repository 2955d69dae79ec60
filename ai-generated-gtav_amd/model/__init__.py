"""Mirrors the reference package `model/` (model/dit.py, model/vae.py)."""
