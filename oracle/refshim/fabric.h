/*
 * fabric.h -- control surface of the in-process NIC (fabric.c) used by the trace driver
 * (refcluster.c).  TEST INFRASTRUCTURE ONLY.
 */
#ifndef APUS_REF_FABRIC_H
#define APUS_REF_FABRIC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct fab_stats {
    uint64_t rc_writes, rc_write_bytes, rc_reads, rc_failures, ud_msgs, ud_dropped;
};

/* called after every successful RDMA WRITE (from_port -> to_port, raw responder address) */
typedef void (*fab_write_hook_t)(void *arg, int from_port, int to_port, uint64_t raddr, uint32_t len);

void fab_reset(void);
/* 1: a send completion is not seen by the poll that directly follows its post (see fabric.c) */
void fab_set_completion_delay(int on);
int  fab_enter(int port);          /* the instance that runs until fab_leave(); returns the previous one */
void fab_leave(int prev);
int  fab_current(void);
void fab_set_write_hook(fab_write_hook_t h, void *arg);
void fab_kill_port(int port);
int  fab_port_alive(int port);
void fab_hold_port(int port);
void fab_release_port(int port);
int  fab_port_held(int port);
int  fab_pending_ud(int port);
const struct fab_stats *fab_get_stats(void);

#ifdef __cplusplus
}
#endif
#endif
