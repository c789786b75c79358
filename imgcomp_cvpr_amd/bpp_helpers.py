"""real bpp + theoretical bpp of one image (mirror of the reference's code/bpp_helpers.py)."""
from . import bit_counter as bc


class BppFetcher(object):
    def __init__(self, pred, checker):
        self.pred = pred
        self.checker = checker

    def get_bpp(self, symbols, num_pixels):
        """:param symbols: NCHW ndarray of one image's symbols -> (bpp_real, bpp_theory)."""
        assert symbols.ndim == 4
        bpp = bc.encode_decode_to_file_ctx(symbols, self.pred, syms_format='CHW', verbose=False) / num_pixels
        bpp_theory = self.checker.get_total_bit_cost(symbols) / num_pixels
        return bpp, bpp_theory


def num_pixels_in_image(im):
    c, h, w = im.shape
    assert c == 3, 'Expected RGB image, got {}'.format(im.shape)
    return w * h
