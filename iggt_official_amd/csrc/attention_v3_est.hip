// The estimated-shift instantiations of flash_attn_d64_v3_kernel (attention_v3.hip, template parameter EST) as their own
// translation unit: see the note at iggt_launch_flash_attn_v3_est there.  Built with the max-ilp scheduling strategy
// (iggt_official_amd/build_ext.py PER_FILE_FLAGS); correct, only slower, without it.
#define IGGT_ATTN_EST_TU 1
#include "attention_v3.hip"
