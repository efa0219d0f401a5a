"""Loader protocol of the reference (data_loaders/base.py:1-11) and the synthetic 'shapes' dataset."""
