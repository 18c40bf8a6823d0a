# debugging aid: rb_sum_blocks24 (permlane swaps + row rotations) replaced by plain butterflies over the lanes with the same lane % 4
SRC = 'resident.hip'
VARIANTS = {
    'wgnaive': [
        ("""__device__ __forceinline__ void rb_sum_blocks24(const float (&v)[24], float (&out)[6]) {
    float s1[12];""", """__device__ __forceinline__ void rb_sum_blocks24(const float (&v)[24], float (&out)[6]) {
    {
        const int lane_ = rs_tid() & 63, row_ = lane_ >> 4, co_ = 2 * (row_ & 1) + (row_ >> 1);
#pragma unroll
        for (int n = 0; n < 6; ++n) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) { float w = v[4 * n + c]; w += __shfl_xor(w, 4, 64); w += __shfl_xor(w, 8, 64); w += __shfl_xor(w, 16, 64); w += __shfl_xor(w, 32, 64); if (c == co_) s = w; }
            out[n] = s;
        }
        return;
    }
    float s1[12];"""),
    ],
}
