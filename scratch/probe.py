import torch, ctypes, os, time
t0=time.time()
lib=ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)),'libprobe.so'))
print('rt version', lib.probe_rtver(), 'torch hip', torch.version.hip)
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0))
x=torch.arange(1000,device='cuda',dtype=torch.float32); y=torch.ones(1000,device='cuda')
s=torch.cuda.current_stream().cuda_stream
lib.probe_axpy.argtypes=[ctypes.c_void_p,ctypes.c_void_p,ctypes.c_float,ctypes.c_int,ctypes.c_void_p]
r=lib.probe_axpy(x.data_ptr(),y.data_ptr(),2.0,1000,s); torch.cuda.synchronize()
print('axpy rc',r,'ok',torch.allclose(y,2*x+1))
st=torch.cuda.Stream()
with torch.cuda.stream(st):
    y2=torch.ones(1000,device='cuda')
    r=lib.probe_axpy(x.data_ptr(),y2.data_ptr(),3.0,1000,torch.cuda.current_stream().cuda_stream)
st.synchronize(); print('side stream ok',torch.allclose(y2,3*x+1))
A=torch.randn(32,2,device='cuda');B=torch.randn(2,32,device='cuda');C=torch.zeros(32,32,device='cuda')
lib.probe_mfma.argtypes=[ctypes.c_void_p]*4
lib.probe_mfma(A.data_ptr(),B.data_ptr(),C.data_ptr(),s); torch.cuda.synchronize()
print('mfma err',(C-A@B).abs().max().item())
import subprocess
print(subprocess.run('nproc; lscpu | grep -E "Model name|^CPU\\(s\\)"; rocm-smi --showmeminfo vram | head -8; ls /dev/dri | head; rocminfo | grep -c gfx950',shell=True,capture_output=True,text=True).stdout)
g=torch.cuda.default_generators[0]; print('gen',g.initial_seed(), g.get_offset()); g.set_offset(8); print(g.get_offset())
print('elapsed',time.time()-t0)
