def save(layer, path, input_spec=None):
    raise NotImplementedError("paddle.jit.save (inference export) is outside the engine's hot path")
