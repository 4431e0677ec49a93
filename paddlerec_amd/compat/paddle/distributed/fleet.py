"""paddle.distributed.fleet (collective mode, tools/trainer.py:113-119): the engine's collective mode is the
row-sharded trainer of paddlerec_amd.trainer (launched with torch.distributed.run); through this namespace only the
single-process dygraph loop is served."""


class DistributedStrategy:
    pass


def init(is_collective=True, strategy=None):
    raise NotImplementedError("runner.use_fleet: run `python -m torch.distributed.run -m paddlerec_amd.trainer ...` "
                              "(row-sharded collective mode) instead")


def distributed_optimizer(optimizer, strategy=None):
    return optimizer


def distributed_model(model):
    return model
