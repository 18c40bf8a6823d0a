# why was the forward prologue 700 us when the step replayed as a hipGraph?  (lesson 42)  variants: no kernel-argument warming; the mailbox word as a
# system-scope atomic load (the version which had the problem); the mailbox word through a scalar load (the version before it)
SRC = 'resident.hip'
_MAIL = "        if (from_mailbox) batch_word = *((const rs_gu64*)&p.mailbox->src[p.x_job] + lz);\n"
VARIANTS = {
    'nowarm': [
        ("    if (rs_wave_of(rs_tid()) == 0) rs_kernarg_warm<(int)(sizeof(dcv_resident_program) + 2 * sizeof(void*))>();\n", ""),
    ],
    'mailatomic': [(_MAIL, "        if (from_mailbox) batch_word = __hip_atomic_load((rs_gu64*)&p.mailbox->src[p.x_job], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);\n")],
    'mailscalar': [(_MAIL, "        if (from_mailbox) batch_word = (rs_u64)p.mailbox->src[p.x_job];\n")],
}
