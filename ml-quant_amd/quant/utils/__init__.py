"""Small utilities (moving average)."""
