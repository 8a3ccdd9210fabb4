"""Reuse of the pre-training backbone downstream: semantic segmentation fine-tuning step."""
