/* The boundary is a C ABI: this file must compile as C99 (no C++), and every entry point must link. */
#include <stdio.h>

#include "../../include/bge_world.h"

int main(void)
{
    bge_world* w = NULL;
    bge_world_desc d = {sizeof(bge_world_desc), -1, NULL, 0};
    bge_world_info info;
    bge_trigger_event ev;
    uint32_t parent[4] = {BGE_NO_PARENT, 0, 1, BGE_NO_PARENT}, slot[4], pass[4], rank[4];
    uint8_t level[4];
    uint64_t load[2];
    int rc = bge_flatten_topology(4, parent, NULL, slot, level, pass, &info);
    if (rc != BGE_OK || info.n_roots != 2 || info.max_depth != 2) return 1;
    if (bge_partition_subtrees(4, parent, NULL, 2, rank, load) != BGE_OK || load[0] + load[1] != 4) return 2;
    rc = bge_world_create(&d, &w); /* fails without a GPU: the message must say so */
    if (rc == BGE_OK) {
        bge_world_destroy(w);
    } else if (bge_last_error()[0] == '\0') {
        return 3;
    }
    (void)ev;
    printf("abi ok (version %x, create rc %d)\n", bge_version(), rc);
    return 0;
}
