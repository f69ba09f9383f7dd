"""Placeholder for backbones/peer of the reference (frozen teacher nets, SURVEY section 8f
rank 3 -- out of scope this round).  The package exists because the reference's IResNet
imports it unconditionally (backbones/frb/iresnet.py:127-128)."""
