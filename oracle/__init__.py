"""oracle/ — TEST INFRASTRUCTURE ONLY: CPU restatements of the reference's algorithms (field.py, tape_eval.py,
bingcd_model.py) and the recipe that compiles the reference's own runtime from /root/reference into oracle/_ref
(Makefile, render_fr.py, emit_ref_cpp.py, ref_loop.cpp, ref_build.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import, call, link or execute anything
here — as the checker or the timed CPU baseline, never as the product: circom_amd/ does not import oracle/, and
every computing call of the C ABI fails with CW_EDEVICE when the HIP library or a GPU is missing."""
