/*
 * apus_smr.h -- the reference's two host-side API surfaces, kept source- and
 * ABI-compatible so that the engine drops in under an unmodified Redis/memcached:
 *
 *   B-outer  LD_PRELOAD hook <-> proxy   /root/reference/src/include/rsm-interface.h:12-15
 *   B-inner  proxy <-> SMR core          /root/reference/src/include/dare/dare_server.h:142-160,197-203
 *                                        /root/reference/src/include/dare/dare_sm.h:42-47
 *                                        /root/reference/src/include/dare/message.h:5-22
 *
 * Implemented in apus_amd/host/apus_proxy.c (plain C) on top of include/apus_gpu.h.
 * The structs below are laid out exactly like the reference's (sizes probed in
 * sizeof(dare_server_input_t) == 216).
 */
#ifndef APUS_SMR_H
#define APUS_SMR_H

#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <unistd.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- B-outer: rsm-interface.h:12-15 ---------------------------------------- */
struct proxy_node_t;
struct proxy_node_t *proxy_init(const char *config_path, const char *proxy_log_path);
void proxy_on_read(struct proxy_node_t *proxy, void *buf, ssize_t ret, int fd);
void proxy_on_accept(struct proxy_node_t *proxy, int ret);
void proxy_on_close(struct proxy_node_t *proxy, int fildes);

/* ---- B-inner: callbacks, dare_sm.h:42-47 ----------------------------------- */
typedef void     (*proxy_store_cmd_cb_t)(void *data, void *arg);
typedef void     (*proxy_do_action_cb_t)(uint16_t clt_id, uint8_t type, size_t data_size, void *data, void *arg);
typedef void     (*proxy_create_db_snapshot_cb_t)(void *snapshot, void *arg);
typedef uint32_t (*proxy_get_db_size_cb_t)(void *arg);
typedef int      (*proxy_apply_db_snapshot_cb_t)(void *snapshot, uint32_t size, void *arg);
typedef void     (*proxy_update_state_cb_t)(void *arg);

/* server types, dare_server.h:23-26 */
#define SRV_TYPE_START 1
#define SRV_TYPE_JOIN  2

/* dare_server_input_t, dare_server.h:142-160 (field order and types unchanged) */
struct dare_server_input_t {
    FILE *log;
    char *name;
    char *output;
    uint8_t srv_type;
    uint8_t sm_type;
    uint8_t group_size;
    uint8_t server_idx;
    proxy_do_action_cb_t do_action;
    proxy_store_cmd_cb_t store_cmd;
    proxy_create_db_snapshot_cb_t create_db_snapshot;
    proxy_get_db_size_cb_t get_db_size;
    proxy_apply_db_snapshot_cb_t apply_db_snapshot;
    proxy_update_state_cb_t update_state;
    char config_path[128];
    void *up_para;
};
typedef struct dare_server_input_t dare_server_input_t;

/* dare_server.h:197-203 */
void   *dare_server_init(void *arg);     /* pthread entry; owns and frees *arg */
void    dare_server_shutdown(void);
int     is_leader(void);
uint8_t get_node_id(void);

/* ---- submission queue -------------------------------------------------------- */
/* Replaces the global TAILQ of 87 KB malloc'd nodes + spinlock (message.h:5-22,
 * proxy.c:147-158): one call enqueues an admitted request for the DARE thread.
 * Thread-safe; copies `len` bytes.  Returns 0, or -1 when the queue is full. */
int apus_tailq_push(uint8_t type, uint16_t connection_id, uint64_t req_id, const void *buf, uint16_t len);
/* The reference's own submission queue (src/include/dare/message.h:5-22) is exported under its names and with its
 * layout -- `tailhead` (a TAILQ_HEAD of tailq_entry_t, 87416 bytes each) and `tailq_lock` -- so that the reference's
 * proxy.c (proxy.c:114-158: malloc, fill, TAILQ_INSERT_TAIL under the spinlock) links against this library
 * unchanged; this library's DARE thread drains it like get_tailq_message (dare_ibv_ud.c:780-790) and frees the nodes.
 * Include the reference's message.h for the types; apus_tailq_drain() = one such drain by hand, returns the count. */
int apus_tailq_drain(void);

/* ---- additions of this build (tests, shutdown) ----------------------------------- */
void     apus_proxy_shutdown(struct proxy_node_t *p);       /* stop the DARE thread and wait for it */
uint64_t apus_proxy_highest_rec(struct proxy_node_t *p);    /* proxy->highest_rec (proxy.h:47) as the device publishes it */
int      apus_proxy_failed(void);                           /* 1 once requests were dropped because the log is full */

/* ---- durability side channel (snapshot of a joiner's donor) ------------------------------- */
/* The snapshot the reference ships to a joining server is the BerkeleyDB records back to back
 * (stablestorage_dump_records, src/proxy/proxy.c:300-304) = the stream apus_gpu_store_stream
 * (apus_gpu.h) regenerates from a replica's HBM.  This is the other end: stablestorage_load_records
 * (proxy.c:306-339) -- walk the records (CONNECT / CLOSE 4 bytes; SEND sizeof(proxy_send_msg) = 24 +
 * the u16 at +8, the reference's overlay), hand each to `store` (what store_record
 * gets: pointer + size; may be NULL) and replay it through `do_action` (connection id, action,
 * data.cmd.len, data.cmd.cmd -- both read at the overlay's place, as the reference does).
 * Returns the number of records, -1 on a malformed stream (an action that is none of the three, or a
 * record that runs past `size`; the reference would run off the buffer there). */
int apus_snapshot_replay(const void *buf, uint32_t size,
                         void (*store)(const void *rec, uint32_t n, void *arg),
                         proxy_do_action_cb_t do_action, void *arg);

/* action codes carried in entry->type, proxy.h:10-12 */
#define PROXY_CONNECT 4
#define PROXY_SEND    5
#define PROXY_CLOSE   6

#ifdef __cplusplus
}
#endif
#endif /* APUS_SMR_H */
