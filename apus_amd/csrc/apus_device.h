/*
 * apus_device.h -- device-side data layout and log algebra of the MI355X
 * consensus engine (gfx950 only; wave64).
 *
 * The ring of every replica is byte-identical to dare_log_t.entries
 * (/root/reference/src/include/dare/dare_log.h:77-102): 64-byte entry header
 * (dare_log.h:33-47) followed by the payload, entries unpadded and unaligned,
 * the two wrap rules of log_append_entry (dare_log.h:502-538) and the
 * "end == len means empty" encoding (dare_log.h:158) included.
 *
 * Next to the ring every replica keeps DERIVED structures that are never part
 * of a digest: an entry directory (offset, length per entry slot, written at
 * append time) so that a wavefront can address 64 consecutive entries without
 * walking headers, one ACK bitmask per slot on the leader (bit i = follower i
 * acknowledged = the reference's reply[i] byte, dare_log.h:41), and the apply
 * stream.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define APUS_DEV_MAX_SERVERS 13
#define APUS_HDR 64u

/* hdr words 0..7 are exactly dare_log_t's offsets (dare_log.h:79-94) */
enum {
    H_HEAD = 0, H_APPLY, H_COMMIT, H_END, H_TAIL, H_OLD_END, H_OLD_COMMIT, H_LEN,
    /* derived counters (entry slots) and small per-replica state */
    H_N_END = 8,      /* entries appended / received                    */
    H_N_PERSIST,      /* entries persisted (old_end as a slot)          */
    H_N_COMMIT,       /* entries committed                              */
    H_N_APPLY,        /* entries applied                                */
    H_LAST_IDX,       /* idx of the entry at tail (leader)              */
    H_SID,            /* ctrl_data->sid, dare_server.h:47-66            */
    H_HIGHEST_REC,    /* proxy->highest_rec, proxy.c:263                */
    H_APPLY_HASH,     /* sum of apply_mix over every upcall             */
    H_APPLY_COUNT,    /* number of upcalls                              */
    H_PREV_HEAD,      /* prev_log_entry_head, dare_server.c:71          */
    H_CID_BITMASK,    /* config.cid.bitmask                             */
    H_CID_EPOCH,
    H_HEAD_SLOT,      /* scratch: newest committed HEAD entry seen by apply (slot+1) */
    H_STORE_COUNT,    /* proxy_store_cmd upcalls (persist_new_entries)  */
    H_N_VISIBLE,      /* leader: entries visible to followers / the ACK scan */
    H_APPLY_OFFSETS,  /* 13 words: ctrl_data->apply_offsets[], dare_server.h:137 */
    H_JOINED_AT = H_APPLY_OFFSETS + APUS_DEV_MAX_SERVERS,   /* a joined server: the entries below this slot were appended by other machines (a former holder of its slot included) */
    H_TERM_SLOT0,     /* a leader: the slot of the first entry it appended in ITS term (the blank CONFIG entry of become_leader): entries
                       * below it are inherited from older terms and commit only behind it (apus_replica.h: rep_commit_pass) */
    H_WORDS = H_APPLY_OFFSETS + APUS_DEV_MAX_SERVERS + 3   /* 38 -> padded */
};

struct apus_apply_rec {       /* == apus_apply_t, include/apus_gpu.h */
    uint64_t slot, off, idx;
    uint32_t len;
    uint16_t clt_id;
    uint8_t  type, kind;
};

/* one replica's HBM-resident state */
struct RepDev {
    uint8_t  *ring;        /* log_len bytes (+ slack)                                    */
    uint64_t *hdr;         /* H_WORDS words                                              */
    uint64_t *dir_off;     /* [dir_cap]  byte offset of entry slot s                     */
    uint32_t *dir_len;     /* [dir_cap]  total bytes of entry slot s (64 + cmd.len)      */
    uint32_t *ack;         /* [dir_cap]  ACK bitmask (meaningful on the leader)          */
    apus_apply_rec *apply; /* [dir_cap]  apply stream, indexed by slot                   */
    uint32_t idx;          /* index in the group                                         */
    uint32_t pad;
};

/* request descriptor staged in HBM (16 bytes, one dwordx4 load per lane) */
struct ReqDev {
    uint64_t req_id;
    uint32_t pay16_type;   /* payload offset / 16 in the low 28 bits, entry type in the top 4 */
    uint16_t len;
    uint16_t clt_id;
};

/* batch-level result of the sequencer, consumed by every later kernel of the call */
struct SeqOut {
    uint64_t e0;           /* log end when the batch starts (len if the log was empty)   */
    uint64_t idx0;         /* idx of the first entry if no restart happens               */
    uint64_t w;            /* virtual position of the entry that wrapped                 */
    uint64_t n_end0;       /* slot of the first entry                                    */
    uint64_t term;
    int64_t  kstar;        /* batch index of the entry that wrapped, -1 if none          */
    int64_t  estar;        /* batch index where idx restarts at 1 (empty log), -1 if none*/
    uint32_t stale;        /* 1: the wrapped entry left a stale header at e0+virt(kstar) */
    uint32_t n;            /* entries in the batch                                       */
    uint32_t head_round;   /* 1: a fused prune tick appended a <HEAD> entry right before  */
    uint32_t pad0;
    uint64_t first_fail;   /* commit scan: first slot without a majority                 */
    uint64_t commit_before;/* leader commit offset before this call                      */
    uint64_t n_commit_before;
    /* context of the call's tail kernels (persist + ACK scan, apply), filled by k_sequence
     * so that each of them starts with ONE load round trip instead of a chain */
    uint64_t vis;          /* slots visible to followers / committable once the batch is appended */
    uint64_t scan_lo;      /* first slot the persist + ACK pass has to look at                   */
    uint64_t np[APUS_DEV_MAX_SERVERS];   /* followers' persisted slot count (~0: nothing left to do) */
    /* Followers on this device that had persisted everything before the call: their persist + ACK
     * of the new entries (reply byte in their own ring and in the leader's, ACK bit) is written by
     * k_append_push together with the entry, in the same 16-byte stores -- no extra HBM traffic. */
    uint32_t fuse_mask;
    uint32_t tail_needed;  /* 0: k_persist_commit has nothing to do (every pushed follower fused, quorum reached) */
    /* 1: every replica on this device is in step (all pushed followers fused, a majority among them,
     * no commit or apply backlog, the whole batch visible): each new entry is committed the moment
     * it is written, so k_append_push also produces its apply records (apply_committed_entries) for
     * the leader and the fused followers, from registers -- k_apply's appliers have nothing to do. */
    uint32_t fast;
    uint32_t pad1;
    uint64_t rec_base;     /* *rec_count when the call started (index of its first per-round record) */
};

struct RepBox;                            /* apus_replica.h: a replica's mailbox (uncached, mapped by its peers) */

/* engine-wide device state */
struct EngDev {
    RepDev   rep[APUS_DEV_MAX_SERVERS];   /* indexed by group index; ring == nullptr when not local */
    RepBox  *box[APUS_DEV_MAX_SERVERS];   /* replica kernels: mailbox of replica i (doorbells in, progress of its followers in) */
    uint8_t *ackb[APUS_DEV_MAX_SERVERS];  /* replica kernels: [follower][dir_cap] ACK byte maps of replica i when it leads (uncached) */
    uint32_t group_size;
    uint32_t leader;                      /* group index, 0xFFFFFFFF = none */
    uint32_t reachable;                   /* bitmask of peers the leader can post to */
    uint32_t dir_mask;                    /* dir_cap - 1 */
    uint32_t flags;                       /* apus_cfg_t.flags (APUS_F_*) */
    uint32_t pad_flags;
    uint64_t log_len;
    uint32_t *status;
    uint32_t *ticket;                     /* [8] arrival counters (k_apply: [1]; k_call: pass [2], scan [3], done [4]) */
    uint32_t *tick_lines;                 /* [32 lines x 32 words] k_call: word 0 append arrivals, word 1 the sequencer's flag, word 2 "inputs fetched" */
    /* staged requests */
    const ReqDev   *req;
    const uint16_t *req_len;
    const uint8_t  *arena;
    const uint32_t *round_first;          /* prefix of round sizes, n_rounds + 1 entries */
    const uint32_t *round_bytes;          /* bytes each round appends (64 + len per request), n_rounds entries */
    const uint64_t *round_prefix;         /* staged rounds only: bytes appended by rounds [0, r), n_rounds + 1 entries */
    const uint32_t *round_change;         /* ... and how often the number of requests per round has changed up to round r (n_rounds + 1 entries) */
    /* per-call scratch */
    SeqOut   *seq;
    uint64_t *round_virt;                 /* [max_rounds + 1] exclusive scan of round bytes */
    uint64_t *round_hash;                 /* [2 * max_rounds] fast path: per-round apply-stream sums (leader kind, follower kind) */
    /* per-round record of the leader since the last reset */
    uint64_t *rec_end, *rec_commit;
    uint64_t *rec_count;
    uint64_t  rec_cap;
    /* multi-segment launches (k_step): per-segment counters / hash scratch, state snapshots
     * (uncached), and the two chain counts (32 replicas each, one per cache line) */
    uint32_t *step_lines;                 /* [segs][32 x 32] */
    uint32_t *step_tickets;               /* [segs][32] */
    uint64_t *step_hash;                  /* [segs][2 x 1024] */
    uint64_t *step_snap;                  /* [segs + 1][SNAP_STRIDE] */
    uint64_t *step_rec;                   /* [segs][REC_WORDS] the segments' sequencing records (uncached, self-tagged granules) */
    uint32_t *step_epoch, *step_seq_done; /* [32 x 32] each: bookkeeper / sequencer of segment k done */
    uint64_t *trace;                      /* -DAPUS_TRACE builds: [kernel][64] wall-clock stamps; else nullptr */
};

/* reply[] bytes of an entry (offsets 28..40, MAX_SERVER_COUNT = 13) for the ACK bits in `mask`:
 * w28 = bytes 28..31 (last word of the second 16-byte unit), x32 / y36 / z40 = words of the third unit */
struct ReplyWords { uint32_t w28, x32, y36, z40; };
__host__ __device__ static inline ReplyWords apus_reply_words(uint32_t mask)
{
    ReplyWords r = {0, 0, 0, 0};
    for (uint32_t m = mask & ((1u << APUS_DEV_MAX_SERVERS) - 1); m; m &= m - 1) {
        const uint32_t f = (uint32_t)__builtin_ctz(m);
        if (f < 4) r.w28 |= 1u << (8 * f);
        else if (f < 8) r.x32 |= 1u << (8 * (f - 4));
        else if (f < 12) r.y36 |= 1u << (8 * (f - 8));
        else r.z40 |= 1u;
    }
    return r;
}

/* ---- log algebra (dare_log.h:255-283) ----------------------------------- */
__host__ __device__ static inline uint64_t apus_end_distance(uint64_t end, uint64_t len, uint64_t off)
{
    if (end == len) return 0;
    if (end >= off) return end - off;
    return len - (off - end);
}
__host__ __device__ static inline bool apus_is_larger(uint64_t end, uint64_t len, uint64_t l, uint64_t r)
{
    return apus_end_distance(end, len, l) < apus_end_distance(end, len, r);
}

/* position-dependent mixing of one apply record; the stream hash is the sum */
__host__ __device__ static inline uint64_t apus_apply_mix(uint64_t slot, uint64_t off, uint64_t idx,
                                                          uint32_t len, uint16_t clt_id, uint8_t type, uint8_t kind)
{
    uint64_t x = slot * 0x9E3779B97F4A7C15ull ^ off * 0xC2B2AE3D27D4EB4Full
               ^ idx * 0x165667B19E3779F9ull
               ^ ((uint64_t)len << 32 | (uint64_t)clt_id << 16 | (uint64_t)type << 8 | kind);
    x ^= x >> 31;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 29;
    return x;
}

#ifdef __HIPCC__
/* unaligned 16-byte global access: gfx950 runs in unaligned access mode, hipcc
 * lowers these to one global_load/store_dwordx4 at any byte address */
/* The pointers of the engine descriptor reach the kernels through memory (kernarg segment, LDS
 * copy), so the compiler sees GENERIC pointers and would emit FLAT loads / stores (aperture check,
 * both wait counters).  Everything they point at is device memory: the helpers below say so
 * (address space 1 = global), which gives global_load / global_store. */
#define APUS_GLOBAL __attribute__((address_space(1)))
typedef uint32_t apus_v4 __attribute__((ext_vector_type(4)));
typedef apus_v4 __attribute__((aligned(1))) apus_v4_u;
typedef uint64_t __attribute__((aligned(1))) apus_u64_u;
__device__ static inline uint4 ld16u(const uint8_t *p)
{
    const apus_v4 v = *(const APUS_GLOBAL apus_v4_u *)(uintptr_t)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
/* Log bytes, directory slots and apply records are written once and not read back by the launch
 * that writes them: streaming (non-temporal) stores.  Written back at the end of the kernel instead,
 * the dirty lines of a 200 MB launch cost ~7 us of every k_step launch (profiles/README.md).
 * -DAPUS_NO_NT_STORES: plain stores, for comparison. */
#ifndef APUS_NO_NT_STORES
__device__ static inline void st16u(uint8_t *p, uint4 v) { __builtin_nontemporal_store(apus_v4{v.x, v.y, v.z, v.w}, (APUS_GLOBAL apus_v4_u *)(uintptr_t)p); }
template <typename T> __device__ static inline void gst_nt(T *p, T v) { __builtin_nontemporal_store(v, (APUS_GLOBAL T *)(uintptr_t)p); }
#else
__device__ static inline void st16u(uint8_t *p, uint4 v) { *(APUS_GLOBAL apus_v4_u *)(uintptr_t)p = apus_v4{v.x, v.y, v.z, v.w}; }
template <typename T> __device__ static inline void gst_nt(T *p, T v) { *(APUS_GLOBAL T *)(uintptr_t)p = v; }
#endif
__device__ static inline uint64_t ld8u(const uint8_t *p) { return *(const APUS_GLOBAL apus_u64_u *)(uintptr_t)p; }
/* naturally aligned words / records in device memory */
template <typename T> __device__ static inline void gst(T *p, T v) { *(APUS_GLOBAL T *)(uintptr_t)p = v; }
template <typename T> __device__ static inline T gld(const T *p) { return *(const APUS_GLOBAL T *)(uintptr_t)p; }

/* physical placement of batch entry gk whose virtual start is a = e0 + virt */
__device__ static inline uint64_t apus_place(const SeqOut &s, int64_t gk, uint64_t a)
{
    if (s.kstar < 0 || gk < s.kstar) return a;
    if (gk == s.kstar) return 0;
    return a - s.w;
}
__device__ static inline uint64_t apus_entry_idx(const SeqOut &s, int64_t gk)
{
    if (s.estar < 0 || gk < s.estar) return s.idx0 + (uint64_t)gk;
    return 1 + (uint64_t)(gk - s.estar);
}
#endif
