# where the prologues of the two persistent launches spend their time: sub-stamps in the backward prologue (positions shift the names printed by
# tools/resident_stamps.py: read the first five rows as issue / loads + loss arithmetic / barrier + writer / commits), and a forward launch which does not
# read the batch address from the host mailbox (wrong on purpose: it reads the static tensor)
SRC = 'resident.hip'
VARIANTS = {
    'bwdsub': [
        ("        if (stage_dz) {\n            float* staged = reinterpret_cast<float*>(dzt);\n", "        stamp();\n        if (stage_dz) {\n            float* staged = reinterpret_cast<float*>(dzt);\n"),
        ("                l = dcv_wave_sum(l); ok = dcv_wave_sum(ok);\n", "                stamp();\n                l = dcv_wave_sum(l); ok = dcv_wave_sum(ok);\n"),
        ("        if (y_two) {\n            float4* d = reinterpret_cast<float4*>(yreg + s_of_t * RS_Y);\n", "        stamp();\n        if (y_two) {\n            float4* d = reinterpret_cast<float4*>(yreg + s_of_t * RS_Y);\n"),
    ],
    'nofence': [   # the loss publication of the backward launch without its two system-scope fences (unordered on purpose: what do they cost the step?)
        ("                            __threadfence_system();\n", ""),
    ],
    'nomail': [
        ("        if (from_mailbox) batch_word = __hip_atomic_load((rs_gu64*)&p.mailbox->src[p.x_job], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);\n", "        if (from_mailbox) batch_word = (rs_u64)o.x;\n"),
    ],
}
