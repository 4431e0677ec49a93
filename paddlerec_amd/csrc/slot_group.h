// Slot-local SelectedRows grouping (gfx950): entry points of csrc/ids_group_slots.hip for sparse_update.hip.
#pragma once
#include "rec_common.h"

namespace rec {
namespace sg {

// true when rec_ids_group_slots takes the slot-local path for this shape (else: the general radix grouping with
// keys id + slot * slot_rows)
bool eligible(int64_t batch, int32_t num_slots, int64_t slot_rows);
size_t workspace_bytes(int64_t batch, int32_t num_slots, int64_t slot_rows);
int run(int64_t batch, int32_t num_slots, int64_t slot_rows, int64_t padding_idx, const int64_t* ids,
        int32_t* sorted_pos, int64_t* uniq_rows, int32_t* seg_offset, int32_t* n_uniq, int32_t* rank,
        int32_t* status, void* workspace, size_t workspace_bytes, hipStream_t st);

}  // namespace sg
}  // namespace rec
