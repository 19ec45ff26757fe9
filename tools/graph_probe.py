import sys, time, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, fep_testlib as tl
pkg=tl.pkg
case = tl.make_case(nm=(40,40,20), num_perturbed_molecules=16, elec="ewald", seed=2026, n_lambda=11, max_cjpacked_per_sci=16)
nb = tl.setup_gpu(case, fused=True, use_dynamic_pruning=True)
sw = pkg.step_workload(energy=False, virial=False, dhdl=False)
for _ in range(20):
    nb.clear_outputs(False); nb.launch_kernel(sw)
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(200):
    nb.clear_outputs(False); nb.launch_kernel(sw)
torch.cuda.synchronize()
print("plain launches: %.4f ms/step" % (1e3*(time.perf_counter()-t)/200))
lib=pkg.hip_lib()
lib.nbnxm_gpu_debug_graph_steps(C.c_void_p(nb._h), C.byref(sw), 5)
t=time.perf_counter()
lib.nbnxm_gpu_debug_graph_steps(C.c_void_p(nb._h), C.byref(sw), 200)
print("graph replay (incl. capture): %.4f ms/step" % (1e3*(time.perf_counter()-t)/200))
t=time.perf_counter()
lib.nbnxm_gpu_debug_graph_steps(C.c_void_p(nb._h), C.byref(sw), 2000)
print("graph replay 2000 (incl. capture): %.4f ms/step" % (1e3*(time.perf_counter()-t)/2000))
