"""paddle.distributed.ShowClickEntry (slot_dnn/net.py:61-62): names the show / click variables the PS accessor reads
when it pushes a feature's statistics [EXT]; it does not change the forward value."""


class ShowClickEntry:
    def __init__(self, show_name, click_name):
        self.show_name, self.click_name = show_name, click_name
