python -m pytest tests -m gpu -x -q -k "bucket_index or get_lower_index or locator" 2>&1 | tail -8
