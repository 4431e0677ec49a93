class ParamAttr:
    def __init__(self, name=None, initializer=None, regularizer=None, learning_rate=1.0, trainable=True):
        self.name, self.initializer, self.regularizer = name, initializer, regularizer
        self.learning_rate, self.trainable = learning_rate, trainable
