import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch, ffi
from ffmpeg_amd import swscale as S
import test_gpu_sws as T
w,h=64,16
rng=np.random.default_rng(w+h)
src=ffi.alloc_frame(0,w,h,rng)
want=T._oracle_unscaled(src,w,h,False)
ctx=S.SwsContext(w,h,0,w,h,2,4)
dsrc=T._upload(src); ddst=[torch.zeros((1,h,3*w),dtype=torch.uint8,device='cuda:0')]
ctx.scale_batch(dsrc,ddst); torch.cuda.synchronize()
got=ddst[0][0].cpu().numpy()
bad=np.argwhere(got!=want)
print(len(bad)); print(bad[:40].tolist())
for y,x in bad[:20]: print(y,x,x//3,x%3,got[y,x],want[y,x], src[0][y,x//3], src[1][y//2,x//6], src[2][y//2,x//6])
