import torch, numpy as np, time, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import tap_clip_amd
from tap_clip_amd import engine
rng = np.random.default_rng(0)
imgs = [torch.from_numpy(rng.integers(0,256,(375,500,3),dtype=np.uint8)).cuda() for _ in range(256)]
for size in (224,):
    out = engine.preprocess_u8(imgs, size=size)
    torch.cuda.synchronize()
    # kernels only: pre-built buffers
    import ctypes as C
    from tap_clip_amd import _lib
    pixels = torch.cat([i.view(-1) for i in imgs]); 
    desc = torch.tensor([(i*375*500*3,375,500,i*375*size*3) for i in range(256)],dtype=torch.int64).cuda()
    ws = torch.empty(256*375*size*3,dtype=torch.uint8,device='cuda'); o = torch.empty(256,3,size,size,device='cuda')
    ms = (C.c_float*6)(*engine.CLIP_MEAN,*engine.CLIP_STD)
    lib=_lib.load()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): lib.tapclip_preprocess_u8(pixels.data_ptr(),desc.data_ptr(),256,size,ms,ws.data_ptr(),o.data_ptr(),st)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.tapclip_preprocess_u8(pixels.data_ptr(),desc.data_ptr(),256,size,ms,ws.data_ptr(),o.data_ptr(),st)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1)/20
    print(f"size {size}: {t*1e3:.1f} us per 256 images ({256/t*1e3:.0f} img/s); equal {torch.equal(o,out)}")
    nbytes = 256*(375*500*3 + 2*375*size*3 + size*size*3*4)
    print(f"algorithmic bytes {nbytes/1e6:.1f} MB -> {nbytes/t/1e6:.1f} GB/s")
