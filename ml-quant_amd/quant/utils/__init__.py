"""Small utilities (moving average, checkpoint files)."""
