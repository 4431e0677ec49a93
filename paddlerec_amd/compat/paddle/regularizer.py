class L2Decay:
    def __init__(self, coeff=0.0):
        self.coeff = coeff
