/* Minimal C client of the counting engine: nothing but include/plastid_counts.h and the shared
 * library -- no Python, no torch.  Counts the 5' ends of five reads over one segment and prints
 * the vector, the way reference plastid's `ga[GenomicSegment("chrA", 95, 125, "+")]` would
 * (genome_array.py:861-928 with FivePrimeMapFactory(offset=2)).
 *
 *   gcc -std=c99 -I include examples/c_client.c -L plastid_amd -lplastid_counts \
 *       -Wl,-rpath,$PWD/plastid_amd -o /tmp/c_client && /tmp/c_client
 */
#include <stdio.h>
#include <stdlib.h>

#include "plastid_counts.h"

#define CHECK(call)                                                            \
    do {                                                                       \
        int rc_ = (call);                                                      \
        if (rc_ != PC_OK) {                                                    \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, pc_last_error()); \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main(void) {
    /* five reads on contig 0, coordinate sorted; the fourth is spliced (10M 30N 15M), the fifth reverse */
    const int32_t tid[5] = {0, 0, 0, 0, 0};
    const int32_t pos[5] = {100, 100, 103, 105, 110};
    const uint16_t alen[5] = {30, 28, 30, 25, 29};
    const uint8_t flags[5] = {0, 0, 0, 0, PC_FLAG_REVERSE};
    const uint8_t nblk[5] = {1, 1, 1, 2, 1};
    const int32_t blk_start[2] = {105, 145}, blk_len[2] = {10, 15};

    pc_engine *eng = NULL;
    CHECK(pc_create(0, &eng));
    CHECK(pc_add_alignment_file(eng, 5, /*ntid=*/1, tid, pos, alen, flags, nblk, 2, blk_start, blk_len));
    CHECK(pc_set_mapping(eng, PC_MAP_FIVE, /*offset=*/2, NULL, NULL, 0, 0, 0));

    /* one '+' segment [95, 125), laid out forward from element 0 */
    const int32_t seg_tid[1] = {0};
    const int64_t seg_start[1] = {95}, seg_end[1] = {125}, out_off[1] = {0}, row_stride[1] = {30};
    const uint8_t strand[1] = {PC_STRAND_FWD};
    const int8_t out_step[1] = {1};
    pc_plan *plan = NULL;
    CHECK(pc_plan_create(eng, 1, seg_tid, seg_start, seg_end, strand, out_off, out_step, row_stride, 30, 1, &plan));
    CHECK(pc_count(eng, plan, PC_OUT_INT64));
    int64_t counts[30];
    CHECK(pc_read_counts(eng, plan, counts, 30));
    int64_t total = 0;
    CHECK(pc_total(eng, plan, &total));

    printf("counts[95..125) =");
    for (int i = 0; i < 30; ++i) printf(" %lld", (long long)counts[i]);
    printf("\ntotal = %lld (forward reads mapped at pos + 2: 102 x2, 105, 107)\n", (long long)total);

    CHECK(pc_plan_destroy(plan));
    CHECK(pc_destroy(eng));
    return 0;
}
