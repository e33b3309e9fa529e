"""Instruction ceilings on this box: a loop of nothing but v_mfma_f32_32x32x2_f32 vs nothing but v_mfma_f32_32x32x16_f16
(profiles/probe/mfma_ceiling.hip), with the shader clock each kernel measures on itself.  Input to DESIGN.md section 9, item 2:
what a split-fp16 (hi/lo, three products) fp32-accurate contraction could reach."""
import ctypes, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.zeros(1, device="cuda")
lib = ctypes.CDLL(os.path.join(ROOT, "profiles", "probe", "libmfma_probe.so"))
out = {}
for name in ("probe_mfma_f32_ceiling", "probe_mfma_f16_ceiling"):
    fn = getattr(lib, name)
    fn.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_void_p]
    v = (ctypes.c_double * 2)(0.0, 0.0)
    rc = fn(v, 5, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    out[name] = {"rc": rc, "TFLOPs": round(v[0], 1), "shader_clock_GHz": round(v[1], 3)}
f32, f16 = out["probe_mfma_f32_ceiling"]["TFLOPs"], out["probe_mfma_f16_ceiling"]["TFLOPs"]
out["fp32_equivalent_of_three_f16_products"] = round(f16 / 3, 1)
out["ratio_to_the_fp32_instruction"] = round(f16 / 3 / f32, 2) if f32 else None
print(json.dumps(out))
