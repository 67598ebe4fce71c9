"""Data side of the hot path: log-mel front-end on the device, text -> ids, batch layout (datasets/ of the reference)."""
