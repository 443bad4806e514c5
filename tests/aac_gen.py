"""Synthetic decoder state for the AAC stereo tools and long-term prediction tests: the IndividualChannelStream /
ChannelElement / LongTermPrediction fields the AACDecDSP members read (libavcodec/aac/aacdec.h), as decode_ics / decode_cpe
(libavcodec/aac/aacdec.c) leave them.  Band offsets are the 44.1 / 48 kHz tables of the standard (ff_swb_offset_1024_48 /
ff_swb_offset_128_48, libavcodec/aactab.c)."""
import numpy as np

SWB_1024 = np.array([0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 72, 80, 88, 96, 108, 120, 132, 144, 160, 176, 196, 216, 240,
                     264, 292, 320, 352, 384, 416, 448, 480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800, 832, 864, 896, 928,
                     1024], np.uint16)
SWB_128 = np.array([0, 4, 8, 12, 16, 20, 28, 36, 44, 56, 68, 80, 96, 112, 128], np.uint16)
NOISE_BT, INTENSITY_BT2, INTENSITY_BT = 13, 14, 15
ONLY_LONG, LONG_START, EIGHT_SHORT, LONG_STOP = 0, 1, 2, 3


def ics(rng, short):
    """window grouping + band limits of one frame"""
    if short:
        cuts = np.flatnonzero(rng.random(7) < .4) + 1
        edges = np.concatenate(([0], cuts, [8]))
        group_len = np.diff(edges).astype(np.uint8)
        swb, nswb = SWB_128, 14
    else:
        group_len, swb, nswb = np.array([1], np.uint8), SWB_1024, 49
    gl = np.zeros(8, np.uint8)
    gl[:len(group_len)] = group_len
    return dict(num_window_groups=len(group_len), group_len=gl, swb=swb, num_swb=nswb, max_sfb=int(rng.integers(1, nswb + 1)))


def cpe(rng, short):
    """a channel pair: common window, ms_mask, band types (both channels), the second channel's scalefactors"""
    c = ics(rng, short)
    n = c["num_window_groups"] * c["max_sfb"]
    c["ms_present"] = int(rng.integers(0, 3))
    c["ms_mask"] = np.zeros(128, np.uint8)
    c["ms_mask"][:n] = 1 if c["ms_present"] == 2 else rng.integers(0, 2, n) * (c["ms_present"] == 1)
    bt0, bt1 = np.zeros(128, np.int32), np.zeros(128, np.int32)
    bt0[:n] = rng.choice([0, 1, 5, 11, NOISE_BT], n, p=[.1, .3, .3, .2, .1])
    bt1[:n] = rng.choice([0, 3, 11, NOISE_BT, INTENSITY_BT2, INTENSITY_BT], n, p=[.1, .25, .2, .1, .15, .2])
    c["band_type0"], c["band_type1"] = bt0, bt1
    sf = np.zeros(128, np.float32)
    sf[:n] = (2.0 ** (rng.integers(-60, 60, n) / 4.0)).astype(np.float32)       # intensity positions -> 0.5^(pos/4)
    c["sf1"] = sf
    return c


def spectrum(rng):
    return (rng.standard_normal(1024) * 10.0 ** float(rng.integers(-1, 4))).astype(np.float32)


def ltp(rng, seq0=None):
    """LongTermPrediction of a long-window frame + the window state windowing_and_mdct_ltp reads"""
    seq0 = int(rng.choice([ONLY_LONG, LONG_START, LONG_STOP])) if seq0 is None else seq0
    c = ics(rng, False)
    c["seq"] = np.array([seq0, int(rng.integers(0, 4))], np.int32)
    c["kb"] = rng.integers(0, 2, 2).astype(np.int32)
    c["lag"] = int(rng.integers(0, 2048))                     # 11 bits in the bitstream
    c["coef"] = float(np.float32([0.570829, 0.696616, 0.813004, 0.911304, 0.984900, 1.067894, 1.194601, 1.369533][int(rng.integers(0, 8))]))
    used = np.zeros(40, np.int8)
    used[:min(c["max_sfb"], 40)] = rng.integers(0, 2, min(c["max_sfb"], 40))
    c["used"] = used
    c["ltp_state"] = (rng.standard_normal(3072) * 3000).astype(np.float32)
    return c
