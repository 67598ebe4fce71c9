"""train / say drivers (run/train.py, run/say.py of the reference) on the HIP engine."""
