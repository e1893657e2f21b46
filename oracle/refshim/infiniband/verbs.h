/*
 * infiniband/verbs.h -- in-process stand-in for libibverbs.  TEST INFRASTRUCTURE ONLY.
 *
 * libibverbs is not installed in this image, and the reference's consensus loops
 * (src/dare/dare_server.c, dare_ibv_rc.c, dare_ibv_ud.c, dare_ibv.c under
 * /root/reference) include this header transitively (src/include/dare/dare_ibv.h:13).
 * This file declares the 27 entry points and the structs/enums those UNMODIFIED sources
 * use; oracle/refshim/fabric.c implements them as an in-process "NIC":
 *   - RC RDMA WRITE/READ = memcpy between N server instances living in one process,
 *     gated by the responder QP's state (a QP in RESET rejects incoming operations --
 *     that is how rc_revoke_log_access, dare_ibv_rc.c:2156, fences a deposed leader);
 *   - UD SEND / multicast = copy into a posted receive buffer behind a 40-byte GRH.
 * Nothing here is derived from rdma-core sources; names and semantics follow the
 * public verbs API that the reference calls.
 */
#ifndef APUS_FAKE_INFINIBAND_VERBS_H
#define APUS_FAKE_INFINIBAND_VERBS_H

#include <stdint.h>
#include <stddef.h>
#include <errno.h>

#ifdef __cplusplus
extern "C" {
#endif

union ibv_gid {
    uint8_t raw[16];
    struct { uint64_t subnet_prefix; uint64_t interface_id; } global;
};

enum ibv_mtu { IBV_MTU_256 = 1, IBV_MTU_512 = 2, IBV_MTU_1024 = 3, IBV_MTU_2048 = 4, IBV_MTU_4096 = 5 };
enum ibv_port_state { IBV_PORT_NOP = 0, IBV_PORT_DOWN = 1, IBV_PORT_INIT = 2, IBV_PORT_ARMED = 3, IBV_PORT_ACTIVE = 4 };
enum { IBV_LINK_LAYER_UNSPECIFIED = 0, IBV_LINK_LAYER_INFINIBAND = 1, IBV_LINK_LAYER_ETHERNET = 2, IBV_LINK_LAYER_SCIF = 3 };
enum ibv_atomic_cap { IBV_ATOMIC_NONE = 0, IBV_ATOMIC_HCA = 1, IBV_ATOMIC_GLOB = 2 };

enum ibv_access_flags {
    IBV_ACCESS_LOCAL_WRITE = 1, IBV_ACCESS_REMOTE_WRITE = 2, IBV_ACCESS_REMOTE_READ = 4,
    IBV_ACCESS_REMOTE_ATOMIC = 8
};

enum ibv_qp_type { IBV_QPT_RC = 2, IBV_QPT_UC = 3, IBV_QPT_UD = 4 };
enum ibv_qp_state { IBV_QPS_RESET = 0, IBV_QPS_INIT, IBV_QPS_RTR, IBV_QPS_RTS, IBV_QPS_SQD, IBV_QPS_SQE, IBV_QPS_ERR };

enum ibv_qp_attr_mask {
    IBV_QP_STATE = 1 << 0, IBV_QP_CUR_STATE = 1 << 1, IBV_QP_EN_SQD_ASYNC_NOTIFY = 1 << 2,
    IBV_QP_ACCESS_FLAGS = 1 << 3, IBV_QP_PKEY_INDEX = 1 << 4, IBV_QP_PORT = 1 << 5,
    IBV_QP_QKEY = 1 << 6, IBV_QP_AV = 1 << 7, IBV_QP_PATH_MTU = 1 << 8, IBV_QP_TIMEOUT = 1 << 9,
    IBV_QP_RETRY_CNT = 1 << 10, IBV_QP_RNR_RETRY = 1 << 11, IBV_QP_RQ_PSN = 1 << 12,
    IBV_QP_MAX_QP_RD_ATOMIC = 1 << 13, IBV_QP_ALT_PATH = 1 << 14, IBV_QP_MIN_RNR_TIMER = 1 << 15,
    IBV_QP_SQ_PSN = 1 << 16, IBV_QP_MAX_DEST_RD_ATOMIC = 1 << 17, IBV_QP_PATH_MIG_STATE = 1 << 18,
    IBV_QP_CAP = 1 << 19, IBV_QP_DEST_QPN = 1 << 20
};

enum ibv_wr_opcode { IBV_WR_RDMA_WRITE = 0, IBV_WR_RDMA_WRITE_WITH_IMM, IBV_WR_SEND, IBV_WR_SEND_WITH_IMM,
                     IBV_WR_RDMA_READ, IBV_WR_ATOMIC_CMP_AND_SWP, IBV_WR_ATOMIC_FETCH_AND_ADD };
enum ibv_send_flags { IBV_SEND_FENCE = 1, IBV_SEND_SIGNALED = 2, IBV_SEND_SOLICITED = 4, IBV_SEND_INLINE = 8 };

enum ibv_wc_status {
    IBV_WC_SUCCESS = 0, IBV_WC_LOC_LEN_ERR, IBV_WC_LOC_QP_OP_ERR, IBV_WC_LOC_EEC_OP_ERR,
    IBV_WC_LOC_PROT_ERR, IBV_WC_WR_FLUSH_ERR, IBV_WC_MW_BIND_ERR, IBV_WC_BAD_RESP_ERR,
    IBV_WC_LOC_ACCESS_ERR, IBV_WC_REM_INV_REQ_ERR, IBV_WC_REM_ACCESS_ERR, IBV_WC_REM_OP_ERR,
    IBV_WC_RETRY_EXC_ERR, IBV_WC_RNR_RETRY_EXC_ERR, IBV_WC_LOC_RDD_VIOL_ERR,
    IBV_WC_REM_INV_RD_REQ_ERR, IBV_WC_REM_ABORT_ERR, IBV_WC_INV_EECN_ERR,
    IBV_WC_INV_EEC_STATE_ERR, IBV_WC_FATAL_ERR, IBV_WC_RESP_TIMEOUT_ERR, IBV_WC_GENERAL_ERR
};
enum ibv_wc_opcode { IBV_WC_SEND = 0, IBV_WC_RDMA_WRITE, IBV_WC_RDMA_READ, IBV_WC_COMP_SWAP,
                     IBV_WC_FETCH_ADD, IBV_WC_BIND_MW, IBV_WC_RECV = 1 << 7, IBV_WC_RECV_RDMA_WITH_IMM };

struct ibv_device   { char name[64]; int fab_port; };
struct ibv_context  { struct ibv_device *device; int fab_port; };
struct ibv_pd       { struct ibv_context *context; uint32_t handle; };
struct ibv_srq;
struct ibv_comp_channel;

struct ibv_device_attr {
    char     fw_ver[64];
    uint64_t max_mr_size;
    int      max_qp, max_qp_wr, max_sge, max_cq, max_cqe, max_mr, max_pd;
    int      max_qp_rd_atom, max_res_rd_atom, max_qp_init_rd_atom;
    enum ibv_atomic_cap atomic_cap;
    int      max_mcast_grp, max_mcast_qp_attach, max_ah, max_srq, max_srq_wr;
    uint16_t max_pkeys;
    uint8_t  phys_port_cnt;
};

struct ibv_port_attr {
    enum ibv_port_state state;
    enum ibv_mtu max_mtu, active_mtu;
    int      gid_tbl_len;
    uint32_t port_cap_flags, max_msg_sz;
    uint16_t pkey_tbl_len, lid, sm_lid;
    uint8_t  lmc, max_vl_num, sm_sl, subnet_timeout, init_type_reply, active_width, active_speed,
             phys_state, link_layer;
};

struct ibv_mr { struct ibv_context *context; struct ibv_pd *pd; void *addr; size_t length;
                uint32_t handle, lkey, rkey; };

struct ibv_cq { struct ibv_context *context; void *cq_context; int cqe; void *fab; };

struct ibv_global_route { union ibv_gid dgid; uint32_t flow_label; uint8_t sgid_index, hop_limit, traffic_class; };
struct ibv_ah_attr { struct ibv_global_route grh; uint16_t dlid; uint8_t sl, src_path_bits, static_rate,
                     is_global, port_num; };
struct ibv_ah { struct ibv_context *context; struct ibv_pd *pd; struct ibv_ah_attr attr; };

struct ibv_qp_cap { uint32_t max_send_wr, max_recv_wr, max_send_sge, max_recv_sge, max_inline_data; };
struct ibv_qp_init_attr {
    void *qp_context; struct ibv_cq *send_cq, *recv_cq; struct ibv_srq *srq;
    struct ibv_qp_cap cap; enum ibv_qp_type qp_type; int sq_sig_all;
};
struct ibv_qp_attr {
    enum ibv_qp_state qp_state, cur_qp_state;
    enum ibv_mtu path_mtu;
    int      path_mig_state;
    uint32_t qkey, rq_psn, sq_psn, dest_qp_num;
    int      qp_access_flags;
    struct ibv_qp_cap cap;
    struct ibv_ah_attr ah_attr, alt_ah_attr;
    uint16_t pkey_index, alt_pkey_index;
    uint8_t  en_sqd_async_notify, sq_draining, max_rd_atomic, max_dest_rd_atomic, min_rnr_timer,
             port_num, timeout, retry_cnt, rnr_retry, alt_port_num, alt_timeout;
};
struct ibv_qp { struct ibv_context *context; void *qp_context; struct ibv_pd *pd;
                struct ibv_cq *send_cq, *recv_cq; struct ibv_srq *srq; uint32_t handle, qp_num;
                enum ibv_qp_state state; enum ibv_qp_type qp_type; void *fab; };

struct ibv_sge { uint64_t addr; uint32_t length, lkey; };
struct ibv_send_wr {
    uint64_t wr_id; struct ibv_send_wr *next; struct ibv_sge *sg_list; int num_sge;
    enum ibv_wr_opcode opcode; int send_flags; uint32_t imm_data;
    union {
        struct { uint64_t remote_addr; uint32_t rkey; } rdma;
        struct { uint64_t remote_addr, compare_add, swap; uint32_t rkey; } atomic;
        struct { struct ibv_ah *ah; uint32_t remote_qpn, remote_qkey; } ud;
    } wr;
};
struct ibv_recv_wr { uint64_t wr_id; struct ibv_recv_wr *next; struct ibv_sge *sg_list; int num_sge; };

struct ibv_wc {
    uint64_t wr_id; enum ibv_wc_status status; enum ibv_wc_opcode opcode; uint32_t vendor_err, byte_len,
    imm_data, qp_num, src_qp; int wc_flags; uint16_t pkey_index, slid; uint8_t sl, dlid_path_bits;
};

struct ibv_device **ibv_get_device_list(int *num_devices);
void ibv_free_device_list(struct ibv_device **list);
const char *ibv_get_device_name(struct ibv_device *device);
struct ibv_context *ibv_open_device(struct ibv_device *device);
int ibv_close_device(struct ibv_context *context);
int ibv_query_device(struct ibv_context *context, struct ibv_device_attr *attr);
int ibv_query_port(struct ibv_context *context, uint8_t port_num, struct ibv_port_attr *attr);
int ibv_query_pkey(struct ibv_context *context, uint8_t port_num, int index, uint16_t *pkey);
int ibv_query_gid(struct ibv_context *context, uint8_t port_num, int index, union ibv_gid *gid);
struct ibv_pd *ibv_alloc_pd(struct ibv_context *context);
int ibv_dealloc_pd(struct ibv_pd *pd);
struct ibv_mr *ibv_reg_mr(struct ibv_pd *pd, void *addr, size_t length, int access);
int ibv_dereg_mr(struct ibv_mr *mr);
struct ibv_cq *ibv_create_cq(struct ibv_context *context, int cqe, void *cq_context,
                             struct ibv_comp_channel *channel, int comp_vector);
int ibv_destroy_cq(struct ibv_cq *cq);
int ibv_poll_cq(struct ibv_cq *cq, int num_entries, struct ibv_wc *wc);
struct ibv_qp *ibv_create_qp(struct ibv_pd *pd, struct ibv_qp_init_attr *qp_init_attr);
int ibv_destroy_qp(struct ibv_qp *qp);
int ibv_modify_qp(struct ibv_qp *qp, struct ibv_qp_attr *attr, int attr_mask);
int ibv_query_qp(struct ibv_qp *qp, struct ibv_qp_attr *attr, int attr_mask, struct ibv_qp_init_attr *init_attr);
int ibv_post_send(struct ibv_qp *qp, struct ibv_send_wr *wr, struct ibv_send_wr **bad_wr);
int ibv_post_recv(struct ibv_qp *qp, struct ibv_recv_wr *wr, struct ibv_recv_wr **bad_wr);
struct ibv_ah *ibv_create_ah(struct ibv_pd *pd, struct ibv_ah_attr *attr);
int ibv_destroy_ah(struct ibv_ah *ah);
int ibv_attach_mcast(struct ibv_qp *qp, const union ibv_gid *gid, uint16_t lid);
int ibv_detach_mcast(struct ibv_qp *qp, const union ibv_gid *gid, uint16_t lid);
const char *ibv_wc_status_str(enum ibv_wc_status status);

#ifdef __cplusplus
}
#endif
#endif
