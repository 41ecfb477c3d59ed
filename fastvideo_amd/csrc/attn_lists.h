// KV block lists shared by a run of query rows (fvk_attn_tile_lists_bf16: sliding-tile windows in tile-major order): the argument block the
// list modes of attn_pp2.hip (8-wave kernel, rounds 1-2) and scripts/probes/attn_w64.hip (4 waves x 64 rows, round 3, measurement build) take.
#pragma once
#include <stdint.h>

struct fvk_pp2_lists {
    const int32_t* q2k_idx; const int32_t* q2k_num; const int32_t* kv_block_sizes; const int32_t* q_rows_valid;
    int max_kv, n_lists, q_stride, q_sub;  // q_sub = 256-row workgroups per list
    const int32_t* o_rows;                 // optional [Sq]: query row r's output goes to row o_rows[r] of o (negative: dropped)
    int plain_ids;                         // measurement: 1 = hardware workgroup order (no XCD-contiguous deal)
};
