"""RNO2dObserver with the reference surface (libs/models/rno_models.py:12-15)."""
from ...neuralop.models.rno import RNO2d


class RNO2dObserver(RNO2d):
    def __init__(self, modes1, modes2, width, recurrent_index, layer_num=3, pad_amount=None, pad_dim='1'):
        super().__init__(modes1, modes2, width, recurrent_index, layer_num=layer_num, pad_amount=pad_amount,
                         pad_dim=pad_dim)
