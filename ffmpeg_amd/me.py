"""Host-side mirror of the reference's me_cmp table / ff_me_search_esa for the batched device faces."""
from . import _lib

SAD, SATD = 0, 1


def _stream(stream):
    import torch
    return torch.cuda.current_stream().cuda_stream if stream is None else stream


def cmp_batch(kind, width, h, blk1, off1, blk2, off2, stride, out, stream=None):
    """out[i] = me_cmp[kind](blk1 + off1[i], blk2 + off2[i], stride, h); tensors on the device."""
    return _lib.check(_lib.lib().ffhip_me_cmp_batch_dev(kind, width, h, blk1.data_ptr(), off1.data_ptr(), blk2.data_ptr(),
                                                        off2.data_ptr(), stride, out.data_ptr(), off1.numel(),
                                                        _stream(stream)), "ffhip_me_cmp_batch_dev")


def esa_batch(cur, ref, width, height, stride, frame_pitch, nframes, mb_size, search_param, cost_kind, mv_out, cost_out,
              stream=None):
    """Exhaustive search of every mb_size block of `cur` in `ref` (ff_me_search_esa semantics), nframes pairs."""
    return _lib.check(_lib.lib().ffhip_me_esa_batch_dev(cur.data_ptr(), ref.data_ptr(), width, height, stride, frame_pitch,
                                                        nframes, mb_size, search_param, cost_kind, mv_out.data_ptr(),
                                                        cost_out.data_ptr(), _stream(stream)), "ffhip_me_esa_batch_dev")
