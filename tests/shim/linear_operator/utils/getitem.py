_noop_index = slice(None, None, None)
