"""paddle.distributed: rank / world of the torch.distributed launch (one process per GPU), 0 / 1 otherwise."""
import os

from . import fleet  # noqa: F401


def get_rank():
    return int(os.environ.get("RANK", "0"))


def get_world_size():
    return int(os.environ.get("WORLD_SIZE", "1"))


class ShowClickEntry:
    """slot_dnn/net.py:61-62: names the show / click variables the PS accessor reads (rec_ps_push_rows takes them
    as arguments)."""

    def __init__(self, show_name, click_name):
        self.show_name, self.click_name = show_name, click_name
