"""paddle.incubate.distributed.fleet.fleet_util.FleetUtil (tools/static_gpubox_trainer.py:32-33,190-197)."""


class FleetUtil:
    def __init__(self, mode="pslib"):
        self.mode = mode

    def set_zero(self, var_name, scope=None, place=None, param_type="int64"):
        """Zeroes a persistable variable (the AUC bucket statistics at the end of an epoch)."""
        from .... import static
        v = (scope or static.global_scope()).find_var(var_name)
        if v is None:
            raise KeyError("no persistable variable %r" % var_name)
        t = v.get_tensor()
        t._t.zero_()
