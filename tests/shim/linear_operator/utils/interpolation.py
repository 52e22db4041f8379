def left_interp(*args, **kwargs):
    raise NotImplementedError("linear_operator shim: interpolation is off the exact-GP path")


left_t_interp = left_interp
