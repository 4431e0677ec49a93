"""paddle.distributed.fleet.base.role_maker (imported by tools/static_gpubox_trainer.py:26; fleet.init() takes the role
from the environment: TRAINING_ROLE / PADDLE_TRAINER_ID, tools/run_gpubox.sh)."""


class PaddleCloudRoleMaker:
    def __init__(self, is_collective=False, **kwargs):
        self.is_collective = is_collective


class Role:
    WORKER, SERVER = 1, 2
