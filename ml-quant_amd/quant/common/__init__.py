"""Model / loss factories and the batch-sharded inference loop."""
