# where does the forward launch's "L4 apply" phase (3.1 us for a 1 us step) spend its time?  variant: the next layer's B fragments requested BEHIND the
# phase's barrier and stamp instead of in front of them
SRC = 'resident.hip'
VARIANTS = {
    'bfraglate': [
        ("""                rs_load_bfrag(p.op[oi + 1].filt, b4);       // (requested in FRONT of the apply step, the 36 registers slowed the general step's loops from 2.2 to 5.3 us and spill in the constant-extent one)
                rs_barrier();
                stamp();
""", """                rs_barrier();
                stamp();
                rs_load_bfrag(p.op[oi + 1].filt, b4);
"""),
    ],
}
