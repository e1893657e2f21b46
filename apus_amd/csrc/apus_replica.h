/*
 * apus_replica.h -- every replica runs its OWN resident kernel: the reference's one-server-per-machine
 * structure on GPUs (one replica per GPU / process; or several on one device, each with its workgroups).
 *
 *   leader workgroups: the dare_server polling() loop of the LEADER
 *   (/root/reference/src/dare/dare_server.c:1012-1125), pipelined --
 *     sequencer (one wavefront)   get_tailq_message + the placement half of log_append_entry
 *                                 (dare_ibv_ud.c:780-790, dare_log.h:466-558): drains the pinned
 *                                 multi-producer request ring (or staged, device-resident rounds, 256
 *                                 per pass), decides where every round goes (both wrap rules, exact fit,
 *                                 the log-full rule), log_pruning ticks (dare_server.c:1996-2067); one
 *                                 ticket per round, MANY rounds in flight
 *     append wavefronts           the byte half of log_append_entry + R1/R2 of update_remote_logs
 *                                 (dare_ibv_rc.c:1465-1643): the round's bytes into the own ring and,
 *                                 write-through, into every pushed follower's ring -- ONLY the E log
 *                                 bytes -- then one 32-byte round doorbell per follower
 *     committer (one wavefront)   the ACK scan of update_remote_logs (dare_ibv_rc.c:1725-1758): the
 *                                 followers' per-replica ACK byte maps of a window of up to 4096
 *                                 entries, eight entries per lane and load; per entry
 *                                 popcount(acks | self) >= size/2+1 (byte-parallel), per lane the
 *                                 count of trailing ones, __ballot over the lanes that are all ones,
 *                                 count-trailing-ones again = commit prefix; R4 commit doorbell
 *     applier (one wavefront)     the leader's apply (dare_server.c:1815-1974) round by round;
 *                                 highest_rec to the host
 *   follower workgroups: the follower's polling() pass on ITS device --
 *     work wavefronts             poll the round doorbell, read the landed entry headers from the own
 *                                 ring, build directory + apply records LOCALLY, persist_new_entries
 *                                 (dare_server.c:1792-1810), rc_send_entries_reply
 *                                 (dare_ibv_rc.c:1828-1863): reply byte in the own log, R3 = the same
 *                                 byte in the sender's log + its ACK byte in the sender's map
 *     retire wavefront            retires rounds in order (end, old_end, store count), tells the leader
 *                                 how far it persisted
 *     apply wavefront             applies on the commit doorbell, tells the leader how far it applied
 *
 * A dead or slow follower costs its ACK, nothing else: the commit is decided by majority, every wait
 * is bounded, a follower that stops consuming its doorbells is dropped from the push set.
 *
 * Hand-offs.  A dependent memory round trip costs 0.5-3 us on this device, so every serial role does ONE
 * per pass and as much work per pass as the rings allow: whatever one role leaves for another in memory is
 * made of self-tagged 8-byte granules {sequence number + 1 : value} that are valid the moment they are
 * seen (no flag, no second drain on the writer's side, no acquire + second load on the reader's);
 * wavefronts of one workgroup talk through LDS.  Between replicas the words cross processes and (on a
 * multi-GPU node) xGMI: every shared word lives in uncached device memory of the replica that READS it
 * (RepBox, ACK maps), is written with system-scope stores through the HIP-IPC mapping and polled locally;
 * ring bytes are written with write-through 16-byte stores, drained (s_waitcnt vmcnt(0)) before the
 * doorbell, and read with system-scope loads (tools/micro/xproc.hip: 0.64 us one way for a doorbell
 * between two processes' kernels on one MI355X, 1.25 us with a drained payload in front).
 */
#pragma once
#include "apus_persistent.h"

/* Round 5: no per-entry ACK byte.  Rounds 3 and 4 had every follower store one byte per entry into a map in the leader's
 * (uncached) memory next to its cumulative in-order count -- a system-scope one-byte store per entry and follower: a
 * read-modify-write of a memory line each, and across GPUs one xGMI write transaction per entry, which is exactly what the
 * reference does badly (rc_send_entries_reply posts one RDMA WRITE of one byte per entry, dare_ibv_rc.c:1828-1863).  The
 * map told the leader nothing the count does not: a follower acknowledges in order or not at all (a round it declines ends
 * its run in front of that round), so "f acknowledged entry s" IS "s < f's count" -- for the entries of a run, for what was
 * appended before it (an exact-fit round held back, entries that had no majority), and for the ACK words the control-plane
 * kernels read when a run ends without a majority.  -DREP_ACK_BYTES=1 keeps the byte map (A/B measurements). */
#ifndef REP_ACK_BYTES
#define REP_ACK_BYTES 0
#endif

#define RB_CAP    8192u          /* round doorbells in flight per follower                    */
#define RS_CAP    16384u         /* leader: tickets in flight                                 */
#define RQ_CAP    (1u << 16)     /* pinned request slots                                      */
#define RA_CAP    (64u << 20)    /* pinned payload arena (bytes)                              */
#define RC_CAP    256u           /* host command ring                                         */
#ifndef R_WIN
#define R_WIN     2              /* 64-slot windows of the request ring whose SLOT words the sequencer reads per round trip: what is not a full window
                                  * of equally long requests (RepReq.ready_win) goes one round of <= 64 at a time, as many per look.  (Rounds 4 / 5
                                  * read 8 / 32 windows of slot words for the bulk passes: two 32-word register arrays, 219 VGPRs) */
#endif
#ifndef R_SUB
#define R_SUB     8              /* 64-round chunks a serial role handles per memory round trip (round 6: 4 -> 8, the committer and the applier
                                  * took 250 of 256 rounds per look at 7 G entries/s) */
#endif
/* Round 6, the lone request's path (host submit -> highest_rec): every hop below is one dependent memory round trip taken out.
 *   REP_SPEC_PAY   a round from the request ring whose entries have ONE size carries it in its ticket (TK_D0; a pinned bulk pass: the
 *                  record's PR_BPF): the append wavefront asks for the slots' inline payload TOGETHER with their descriptors and
 *                  loads nothing a second time when the descriptors say what the ticket said
 *   REP_APPLY_PRE  the leader's applier, idle, holds the done granules of the next ticket before the commit reaches it
 *   REP_BELL16     a round doorbell is one 128-byte line: granules 8..15 = what emeta[] says of the round's first eight entries
 *                  (a round of <= 8 entries: no second look by the follower's work wavefront)
 *   REP_POLL_WIDE  an append wavefront that has waited this many polls looks at its ticket's eight words at once
 * -D...=0: A/B measurements (profiles/r06_latency_ab.txt). */
#ifndef REP_SPEC_PAY
#define REP_SPEC_PAY 1
#endif
#ifndef REP_APPLY_PRE
#define REP_APPLY_PRE 1
#endif
#ifndef REP_BELL16
#define REP_BELL16 1
#endif
#ifndef REP_FAST_ACK
#define REP_FAST_ACK 1              /* a follower's work wavefront that holds the very round its retire wavefront is waiting for acknowledges it itself */
#endif
#ifndef REP_POLL_WIDE
#define REP_POLL_WIDE 0             /* (measured: 1536 idle wavefronts looking at twelve lines each instead of two cost the lone request 0.2-0.6 us) */
#endif
#define R_HOLD_MAX 2u                /* REP_APPLY_PRE: done tickets in front of the applier up to which it takes them one by one, holding their granules */
#define R_QUIET    32u               /* ... and the idle passes in a row that say "rounds come one by one" */
#define R_BELL_W  (REP_BELL16 ? 16 : 8)      /* granules per round doorbell */
#define GP_MIN    256u           /* a pass of at least this many plain staged rounds is its record ALONE: no word per ticket (below) */
#define GR_CAP    128u           /* such records in flight: RS_CAP / GP_MIN = 64 passes' tickets fill the ticket ring */
#define GP_MAX    4096u          /* staged rounds ONE pass of the sequencer may take (one record; two words per ticket below GP_MIN) */
#define GP_GOAL   1024u          /* ... and the room it waits for when the rings run full: tickets are handed out this many at a time */
#define GP_GRP    8              /* 64-ticket chunks of a pass whose words are loaded together */
#define R_LAT_CAP (1u << 16)
#define R_SLACK   (3u * WAVE)    /* head room kept in the ticket / doorbell rings             */
#define R_INLINE  96u            /* payload bytes that fit into the request slot itself       */
#define R_PAY_INLINE 0x0FFFFFFFu /* ReqDev.pay16_type: the payload sits in the slot           */

enum { R_OP_PRUNE = 1, R_OP_RUN = 2, R_OP_STOP = 3 };
enum { R_SRC_PINNED = 0, R_SRC_STAGED = 1, R_SRC_CONTROL = 2 };
/* exit codes */
enum { R_EXIT_STOP = 0, R_EXIT_IDLE = 1, R_EXIT_TIMEOUT = 2, R_EXIT_GAP = 3, R_EXIT_FENCED = 4 /* a follower met a round of a term older than its own */ };

/* one host command = four self-tagged granules {command number + 1 : value}: op, low half of the request
 * slot count it waits for, a, b */
struct RepCmd { volatile uint64_t g[4]; };
/* one request slot: descriptor + the payload itself when it is short */
struct RepSlot { ReqDev d; uint8_t pay[112]; };

/* host <-> leader: pinned, coherent (hipHostMalloc mapped) */
struct RepHost {
    /* host -> kernel */
    volatile uint64_t stop;
    uint64_t pad0[7];
    /* kernel -> host */
    volatile uint64_t cmd_head;          /* commands carried out                               */
    volatile uint64_t slots_done;        /* request slots whose bytes were read (ring reuse)   */
    volatile uint64_t highest_rec;       /* proxy->highest_rec (src/proxy/proxy.c:263)         */
    volatile uint64_t commit_slot;
    volatile uint64_t alive;             /* 1 running, 2 exited                                */
    volatile uint64_t exit_code;
    volatile uint64_t full;              /* rounds refused: the log was full                   */
    volatile uint64_t rounds;            /* tickets issued (written when the run ends)         */
    volatile uint64_t settled;           /* commands carried out + request slots taken whose rounds are all in every ring, committed and applied as far as a majority allows */
    uint64_t pad1[7];
};
/* host -> leader: what the HOST writes and the leader's kernel reads -- the command ring and the multi-producer request
 * ring.  Round 4: in DEVICE memory where the host can store into it (large BAR: hipDeviceAttributeIsLargeBar): a producer's
 * descriptor, payload and publish word are posted writes across PCIe, the sequencer polls and the append wavefronts read
 * LOCAL memory (round 3: pinned host memory, every poll and both dependent reads of a lone request a PCIe round trip of
 * 2.4 us: tools/micro/bar.hip -- host store -> resident kernel -> host 1.7 us through the BAR against 2.4 us).  Pinned host
 * memory otherwise (APUS_REQ_RING=host forces it).  The host never READS this block. */
struct RepReq {
    volatile uint64_t stop;              /* host -> kernel: leave at the next look (apus_gpu_rep_park when the command ring has no room) */
    uint64_t pad0[7];
    RepCmd   cmd[RC_CAP];
    /* multi-producer request ring: a producer reserves slot (+ arena range for a long payload), copies
     * the payload, fills slot[].d, then publishes ready_len[slot] = tag << 16 | len (release) */
    volatile uint32_t ready_len[RQ_CAP];
    /* Round 6: one word per aligned WINDOW of 64 slots, tag << 16 | len like the slots' own, written by a producer that has
     * published all 64 of them itself with one length (apus_gpu_rep_submit: a block of 256 slots = four windows).  The
     * sequencer looks at 64 of these per round trip -- 4096 slots, one lane per window -- where rounds 4 and 5 looked at R_WIN
     * windows' 64 slot words each (8 -> 32 windows per look bought 167 -> 300 M entries/s and 64 VGPRs of the leader's kernels) */
    volatile uint32_t ready_win[RQ_CAP / WAVE];
    RepSlot  slot[RQ_CAP];
    uint8_t  arena[RA_CAP + 64];
};
__host__ __device__ static inline uint32_t rep_slot_tag(uint64_t slot) { return (uint32_t)((slot / RQ_CAP) % 65535u) + 1u; }

/* one replica's mailbox: uncached device memory of the replica, mapped by its peers */
struct RepBox {
    /* the leader -> this replica.  One round = four self-tagged granules {seq + 1, value}:
     *   [0] end offset after the round   [1] low 32 bits of the slot count after the round
     *   [2] end offset before the round (len: the log read as empty)
     *   [3] n << 17 | T when all n entries are T bytes long, n << 17 when the sizes differ (lens[])
     * (written as a whole 64-byte line: a partial line is a read-modify-write in the memory that receives it) */
    uint64_t rnd[RB_CAP][R_BELL_W];      /* (round 6: sixteen granules, one 128-byte line; [8..15]: the first eight entries' emeta words) */
    uint16_t lens[RB_CAP][WAVE];         /* cmd.len of every entry of a round of mixed sizes (2 B per entry) */
    /* Round 5: what a follower needs to know of every entry of a client round beyond what the doorbell says -- clt_id, type,
     * sender: the third word of the header's second half, 4 B per entry -- so that it does not have to READ the headers that
     * landed in its ring (a 32-byte read costs the 128-byte line: 256 B of the 884 an entry moved at three replicas; without
     * it +13 % / +15 % at five / seven replicas).  Doorbell granules 4..7 carry idx0 and the term (R_BELL_META). */
    uint32_t emeta[RB_CAP][WAVE];
    uint64_t commit_bell;                /* R4: committed slots                                 */
    uint64_t ctrl;                       /* (f_runs + 1) << 40 | rounds of this run to consume + 1: park */
    uint64_t ping, pong;                 /* link calibration (k_calib_pingpong): the word the peer writes, the word it answers in */
    uint64_t pad0[4];
    /* followers -> this replica while it leads (index = follower) */
    uint64_t seqdone_by[16];             /* rounds applied (their doorbell slots are free)      */
    /* R3, cumulative and IN ORDER: (f_runs + 1) << 40 | entry slots follower f holds and has persisted, every one of
     * them -- written by its retire wavefront, which walks the rounds in order.  What the leader's commit is decided
     * from: an entry counts as acknowledged by f only when f holds everything in front of it, as with the reference's
     * in-order RC writes (dare_ibv_rc.c:1828-1863 walks old_end -> end).  Round 3 counted per-round ACK granules sent by
     * the work wavefronts as the rounds landed, out of order: a leader that died between two rounds' arrivals could
     * have committed (and told a client about) a round that no survivor held behind a contiguous log. */
    uint64_t persisted_by[16];
    uint64_t applied_by[16];             /* entry slots applied                                 */
    uint64_t apply_off_by[16];           /* ... and the apply offset that goes with it (when the run ends) */
    uint64_t sid_by[16];                 /* a follower that moved on to a newer SID says so here: the term fence */
    /* Round 6 (REP_FAST_ACK): the same count as persisted_by[], raised (atomic max, system scope) by the follower's WORK wavefront
     * when the round it has just taken in is the one its retire wavefront stands in front of -- every round before it retired in
     * order, this one continues them (RepFollow.ret_pub), every entry acknowledged, not an exact fit: what the retire wavefront
     * will say one hand-off through memory later (~1 us of a lone round's commit latency).  The leader commits on the larger of
     * the two words; persisted_by[] alone still carries every ACK. */
    uint64_t persisted_fast_by[16];
    /* this replica's own notes, kept across runs of its follower workgroups */
    uint64_t f_seq_next;                 /* next round it expects                               */
    uint64_t f_pend_slot0, f_pend_slot_end, f_pend_sid;   /* an exact-fit round it holds back (its end == len) */
    uint64_t f_exit;                     /* exit code + 1 of the last run                       */
    uint64_t f_runs;                     /* runs of its follower workgroups that have ended     */
    uint64_t pad1[2];
};

/* A serial role is ONE wavefront that hundreds wait for, on a SIMD it shares with one or two append / work wavefronts: its
 * instructions go first (s_setprio: the arbiter picks the wavefront with the highest priority that is ready).  -DREP_NO_PRIO: A/B. */
#ifndef REP_NO_PRIO
#define REP_SERIAL_PRIO() __builtin_amdgcn_s_setprio(3)
#else
#define REP_SERIAL_PRIO() ((void)0)
#endif
/* one round, sequencer -> append wavefront: eight words {low 16 bits of ticket + 1 : 48-bit value}, valid the
 * moment all eight carry the ticket's tag -- the sequencer never waits for its stores.  Kept word-major
 * (RepLead.tkw[word][ticket]): the sequencer holds one ticket per lane, so word w of 64 consecutive tickets is ONE
 * store instruction over 512 contiguous bytes -- whole lines, no transposition through LDS in the serial role
 * (round 3 did one: 4 KiB through LDS and back per 64 tickets, two LDS round trips per chunk of a pass). */
enum { TK_E0 = 0, TK_IDX0, TK_SLOT0, TK_SRC, TK_END, TK_D0, TK_D1, TK_META };
/* TK_D1: control entry data word 1, else the low half of the sequencer's wall clock (latency samples)
 * TK_META: [7:0] n  [11:8] source kind  [12] hidden (the round ends exactly on len)  [23:16] control entry type
 *          [47:32] push mask */
/* BULK passes (round 4).  Writing eight words per ticket is what the sequencer's time went into (a wavefront issues one
 * instruction every four to five cycles: 4 us per pass of 256 tickets, 28 ns per round -- the whole kernel's ceiling).  A
 * pass of >= 64 PLAIN staged rounds (no wrap, no exact fit, room in the ring: positions are prefix arithmetic) is therefore
 * ONE pass record {first ticket, first round, end / idx / slot before the pass, the prefix sums of its first round, push
 * mask, clock} plus TWO words per ticket (TK_META with TK_BULK set: the round's number and the low bits of the pass number; TK_SRC: the
 * round's first request and their number, so that the descriptors can be asked for together with the record);
 * the append wavefront works the other seven words out for itself from the record and the staged prefix sums
 * (E.round_prefix / E.round_first: read-only during a run).  Everything else -- pinned rounds, control entries, the round at
 * a wrap, short passes -- keeps its eight words. */
#define TK_BULK   (1ull << 13)          /* TK_META: [11:0] pass number (low bits)  [13] 1  [47:16] staged round; TK_SRC: [31:0] first request  [38:32] n */
#define TK_BULK_PIN (1ull << 14)        /* ... a pass of FULL, equally long rounds from the request ring: [47:16] = the round's number within the pass,
                                         * PR_BPF = the entries' size, PR_BRF_N = the pass's first request slot (low half) | rounds << 32 */
/* WORDLESS passes (round 6).  A pass of >= GP_MIN plain staged rounds -- configs[1]: the 1024 rounds between two prune ticks -- is
 * its record and nothing else: the record goes into a ring of its own (RepLead.grec, numbered by S.g_seq), and an append
 * wavefront FINDS the pass that holds its ticket: it keeps the number of the first such pass it has not left behind, asks for
 * that record and the three after it together with its ticket's words (one load instruction: lanes 0..7 the words, lanes 8..39
 * four records) and steps over the passes whose tickets lie below its own.  A pass has >= GP_MIN tickets, a wavefront's tickets are
 * 4 n_append apart: it never falls more than a few records behind, and never GR_CAP.  With no word written per ticket a slot of
 * tkw[] would keep an OLD ticket's words -- and their tag comes round again after 16384 x 65535 tickets -- so a wavefront that
 * has taken a ticket with words clears the slot's TK_META.  PR_BRF_N of such a record: [31:0] first request  [44:32] rounds
 * [47] every round of the pass has the same number of requests, PR_RC0[39:32] (EngDev.round_change says so: the first request
 * of round r is then first + n (r - rc0), no look at round_first[]).  The sequencer's pass costs the same whatever it takes:
 * one round trip + ~300 instructions. */
#define PR_UNIFORM (1ull << 47)
#define PR_CAP    1024u                 /* pass records (a bulk pass has >= 64 tickets, RS_CAP tickets are in flight at most) */
enum { PR_T0 = 0, PR_RC0, PR_END0, PR_IDX0, PR_SLOT0, PR_BPF, PR_BRF_N, PR_PUSH_STAMP };   /* words {low 16 bits of pass + 1 : 48-bit value} */
#define TK_VAL 0x0000FFFFFFFFFFFFull
/* (the tag is never 0: a word nobody has written yet never reads as valid.  Round 4: the words of a slot that bulk passes
 * do not rewrite can stay zero for the whole run) */
__device__ static inline uint64_t rep_tag16(uint64_t t) { return t % 65535ull + 1ull; }
__device__ static inline uint64_t rep_tk(uint64_t t, uint64_t v) { return (rep_tag16(t) << 48) | (v & TK_VAL); }
__device__ static inline bool rep_tk_ok(uint64_t w, uint64_t t) { return (w >> 48) == rep_tag16(t); }
/* one round as its append wavefront leaves it for the committer / the applier: granules {ticket + 1 : value} */
enum { DN_META = 0, DN_SLOT_END, DN_END, DN_HASH_LO, DN_HASH_HI, DN_NCLIENT, DN_T_APPENDED, DN_T_SEQUENCED };
/* DN_META: [7:0] n  [11:8] source kind  [12] hidden  [31:16] push mask */

/* leader-local state shared by its workgroups (device memory, agent scope) */
struct RepLead {
    uint64_t pub;       uint64_t pad0[7];       /* low half of (commands carried out + request slots taken) << 32 | low half of the tickets issued; the progress is written with the tickets it made */
    uint64_t seq_final; uint64_t pad1[7];       /* ~0 while running, then the number of tickets */
    uint64_t drop_mask, slots_dropped, pad2[6];
    uint64_t t_drop[16];                        /* tickets issued when follower f left the push set (~0: still in) */
    uint32_t lat_n, pad3;
    uint64_t stat[6][8];                        /* per serial role (sequencer, committer, applier): passes, passes that moved something, items, wall-clock ticks; [3] append phase timers; [4] more of the sequencer's */
    uint32_t lat_ticks[R_LAT_CAP];              /* sequenced -> committed and applied by the leader      */
    uint32_t lat_app[R_LAT_CAP];                /* bytes in every pushed ring -> committed and applied   */
    uint64_t  tkw[8][RS_CAP];
    uint64_t  prec[PR_CAP][8];
    uint64_t  grec[GR_CAP][8];                  /* the records of the passes without ticket words, by their own sequence number */
    uint8_t   xcc[1024];                        /* which XCD (XCC_ID) every workgroup of the launch ran on: diagnostics (apus_gpu_rep_xcc_map) */
    uint64_t  dn[8][RS_CAP];                     /* granule-major: the committer / applier read 64 consecutive tickets' granules in whole lines */
};
/* the leader's first workgroup: its wavefronts' words in LDS */
enum { M_TAIL = 0, M_PROG, M_FINAL, M_T_DONE, M_CS, M_C_FINAL, M_T_RETIRED, M_N_APPLY, M_A_FINAL, M_A_HASH, M_A_NCL, M_DROPPED,
       M_WORDS = 16 };

/* one round as a follower's work wavefront leaves it for its retire / apply wavefronts: granules {round + 1 : value} */
enum { FR_END = 0, FR_E0, FR_SLOT_END, FR_N, FR_HASH_LO, FR_HASH_HI, FR_HEAD, FR_WORDS = 8 };
/* FR_N: [7:0] n  [15:8] client entries  [16] every entry acknowledged (none of a term older than this server's)
 * FR_HEAD: the head a <HEAD> entry carries, 0xFFFFFFFF none.
 * FR_END .. FR_N go out as soon as the round's entries are read and its stores are issued (the retire wavefront's
 * cumulative ACK does not wait for this server's own bookkeeping to drain), FR_HASH_LO .. FR_HEAD behind the drain. */
#define PB_VAL 0xFFFFFFFFFFull
struct RepFollow {                       /* follower-local (device memory, agent scope) */
    uint64_t quit; uint64_t pad[7];
    uint64_t stat[3][8];                 /* retire / apply wavefront: passes, passes that moved something, rounds, wall-clock ticks; [2]: the work
                                          * wavefronts' phase timers (APUS_REP_DBG & 256): rounds, total, bell -> headers, headers -> stores issued, drain */
    uint64_t fr[FR_WORDS][RB_CAP];          /* granule-major, like the leader's done granules */
    /* where the retire wavefront stands, for the work wavefronts (REP_FAST_ACK): granules {q_ret + 1 : low half of the end offset /
     * of the slot count every round before q_ret left} -- not valid (0) while an exact-fit round is held back */
    uint64_t ret_pub[8];
};
/* a follower's first workgroup: its retire / apply wavefronts' words in LDS */
enum { F_END = 0, F_N_END, F_Q_RET, F_R_FINAL, F_STORE_COUNT, F_PEND_N, F_PEND_SLOT_END, F_EXIT, F_WORDS = 8 };

/* host <-> one hosted follower: pinned, coherent.  A follower's process follows its replica's progress here while
 * the run is resident (its DARE thread replays what is applied into its own application, proxy.c:341-439) and can
 * ask its workgroups to leave when the leader is gone (nobody rings the park doorbell then). */
struct RepFHost {
    volatile uint64_t stop;              /* host -> kernel: leave at the next look                */
    uint64_t pad0[7];
    volatile uint64_t n_apply;           /* kernel -> host: entry slots applied                   */
    volatile uint64_t n_end;             /*                 entry slots persisted                 */
    volatile uint64_t alive;             /* 1 running, 2 left                                     */
    volatile uint64_t exit_code;
    /* host -> kernel: a host consumer replays the apply stream into its own application (proxy_do_action, proxy.c:341-439:
     * in the reference the apply IS that call).  consumer != 0: what this server tells the leader it has applied
     * (applied_by: the prune samples' verification, and with it how far the head may move) is the SMALLER of the
     * device's apply count and `replayed`, the entry slots the host has carried out -- the log cannot be pruned and lapped
     * under a host replay that lags (round 3: the device count alone). */
    volatile uint64_t consumer;
    volatile uint64_t replayed;
    uint64_t pad1[2];
};

struct RepArgs {
    RepHost *H;                          /* leader here: its pinned block                         */
    RepReq  *RQ;                         /* ... and the command + request rings the host fills    */
    RepFHost *FH[APUS_DEV_MAX_SERVERS];  /* followers hosted here: their pinned blocks            */
    RepLead *LS;
    RepFollow *FS[APUS_DEV_MAX_SERVERS]; /* followers hosted here                                 */
    uint32_t lead_here;                  /* 1: the first 1 + n_append workgroups are the leader's */
    uint32_t push_mask;                  /* followers in step that get every round of this run    */
    uint32_t park_mask;                  /* followers the leader tells to park at the end         */
    uint32_t follow_mask;                /* followers whose workgroups this launch carries         */
    uint32_t n_append, n_fwork;          /* append workgroups; workgroups per follower             */
    uint64_t fruns[APUS_DEV_MAX_SERVERS];/* leader: every follower's f_runs when this run began       */
    uint64_t idle_polls, peer_polls;
    uint64_t qbase[APUS_DEV_MAX_SERVERS];
    uint32_t dbg, pad_dbg;               /* measurements only (APUS_REP_DBG): 1 no reply bytes, 2 no follower directory / apply records, 4 no push */
};

/* a short nap between two polls while work is expected, a longer one once the poller has been idle */
__device__ static inline void rep_nap(bool eager) { if (eager) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(6); }
__device__ static inline uint32_t ld_sys32(const volatile uint32_t *p) { return __hip_atomic_load((const APUS_GLOBAL uint32_t *)(uintptr_t)p, RLX_SYSTEM); }
__device__ static inline void st_sys8(uint8_t *p, uint8_t v) { __hip_atomic_store((APUS_GLOBAL uint8_t *)(uintptr_t)p, v, RLX_SYSTEM); }
__device__ static inline uint8_t ld_sys8(const uint8_t *p) { return __hip_atomic_load((const APUS_GLOBAL uint8_t *)(uintptr_t)p, RLX_SYSTEM); }
__device__ static inline uint8_t rep_ack_tag(uint64_t slot, uint32_t dir_mask)
{
    return (uint8_t)(((slot / ((uint64_t)dir_mask + 1)) & 0x7F) + 1);
}
/* a granule: {sequence number + 1 : 32-bit value} */
__device__ static inline uint64_t rep_gran(uint64_t seq, uint32_t v) { return ((seq + 1) << 32) | v; }
__device__ static inline bool rep_gran_ok(uint64_t g, uint64_t seq) { return (uint32_t)(g >> 32) == (uint32_t)(seq + 1); }
/* a 64-bit counter from its low half and a nearby (not larger by 2^31) value of the same counter */
__device__ static inline uint64_t rep_extend(uint64_t near, uint32_t lo) { return near + (uint64_t)(int64_t)(int32_t)(lo - (uint32_t)near); }
/* 32 bytes of a ring that a peer's kernel wrote (system-scope loads: never an L1 copy of an older lap) */
__device__ static inline void ld32_sys(const uint8_t *p, uint4 &a, uint4 &b)
{
    v4u_t d0, d1;
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(d0), "=&v"(d1) : "v"(p) : "memory");
    a = make_uint4(d0.x, d0.y, d0.z, d0.w); b = make_uint4(d1.x, d1.y, d1.z, d1.w);
}
/* payload_unit() in two halves: the load (issued for every lane, never inside a divergent branch -- a load in a
 * branch is waited for where the branch ends, one memory round trip per unit) and the masking of what came back.
 * 16 bytes [so, so+16) of the byte stream of a client entry, so >= 48: [48,50) cmd.len, [50,50+P) payload, rest 0. */
__device__ static inline uint4 payload_mask(uint4 v, uint32_t so, uint32_t P, uint32_t len16)
{
    uint64_t lo = 0, hi = 0;
    if (P != 0) {
        lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
        hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
        const int vb = so < 50 ? (int)(50 - so) : 0;
        const int ve = (int)min(16u, 50u + P - so);
        lo &= byte_mask64(vb, ve);
        hi &= byte_mask64(vb - 8, ve - 8);
    }
    if (so == 48)      lo |= (uint64_t)(len16 & 0xFFFFu);
    else if (so == 49) lo |= (uint64_t)((len16 >> 8) & 0xFFu);
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
/* 16 bytes, dword aligned, written through (system scope): st16_agent without its byte-wise path for unaligned addresses.
 * (s_nop 1 INSIDE the string: a VMEM store of more than 64 bits reads its data registers a wait state after it issues, and the
 * compiler pads nothing around an asm statement -- a VALU write of those registers straight behind the store changes what is
 * stored.  cdna_hip_programming.md 5.7 item 1 says so; round 5's first-contact test (apus_selftest.h) showed it: its unrolled
 * pattern loop had the first two words of unit u + 64 in unit u, one unit in five.  The data path's own loops never put a VALU
 * write of the data registers there -- parity at full size says so -- but nothing kept the compiler from doing it.) */
#ifndef REP_ST_PAD
#define REP_ST_PAD "\n\ts_nop 1"      /* (-DREP_ST_PAD='""': A/B measurements of the pad only -- never a product build) */
#endif
__device__ static inline void st16_wt(uint8_t *p, uint4 v)
{
    v4u_t d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" REP_ST_PAD :: "v"(p), "v"(d) : "memory");
}
/* 16 bytes as a streaming store: acknowledged by the L2, on its way to memory behind that -- visible to others only
 * behind rep_release() */
__device__ static inline void st16_nt(uint8_t *p, uint4 v)
{
    v4u_t d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off nt" REP_ST_PAD :: "v"(p), "v"(d) : "memory");
}
/* everything this wavefront has stored is in memory (system scope): write-back of the L2's dirty lines + drain */
__device__ static inline void rep_release()
{
    asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_wbl2 sc0 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
}
/* 16 bytes written through to the device's memory (agent scope: sc1) */
__device__ static inline void st16_dev(uint8_t *p, uint4 v)
{
    v4u_t d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" REP_ST_PAD :: "v"(p), "v"(d) : "memory");
}
__device__ static inline void ld32_dev(const uint8_t *p, uint4 &a, uint4 &b)
{
    v4u_t d0, d1;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(d0), "=&v"(d1) : "v"(p) : "memory");
    a = make_uint4(d0.x, d0.y, d0.z, d0.w); b = make_uint4(d1.x, d1.y, d1.z, d1.w);
}
/* lane l's 64-bit value as a scalar (l wave-uniform): v_readlane, no trip through the LDS crossbar */
__device__ static inline uint64_t rdl64(uint64_t v, int l)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
}
/* Crossing lanes.  __shfl / __shfl_xor / __shfl_up compile to ds_bpermute_b32 here: a trip through the LDS crossbar,
 * ~100+ cycles each and a dependent chain of 12 of them for one 64-bit reduction -- a serial role that reduces, scans
 * and picks "the value of the last lane that ..." a dozen times per pass spends its pass on that (round 4: the
 * sequencer's 4.4 us per 256 rounds, the applier's 3 us).  So: a WAVE-UNIFORM lane index reads through v_readlane
 * (rl64u / rl32u), sums / scans / minima over the wavefront go through DPP row shifts + row broadcasts (wscan32 ...),
 * and only a genuinely per-lane index pays for the crossbar (rl64v).  The DPP forms need all 64 lanes active. */
__device__ static inline uint64_t rl64v(uint64_t v, int l)       /* l may differ from lane to lane */
{
    return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), l, WAVE) << 32) | (uint32_t)__shfl((int)(uint32_t)v, l, WAVE);
}
__device__ static inline uint64_t rl64u(uint64_t v, int l)       /* l wave-uniform */
{
    l = __builtin_amdgcn_readfirstlane(l) & (WAVE - 1);
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
}
__device__ static inline uint32_t rl32u(uint32_t v, int l)       /* l wave-uniform */
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(l) & (WAVE - 1));
}
#define REP_DPP(old, v, ctrl, rows) ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(v), ctrl, rows, 0xf, false))
/* inclusive scan over the 64 lanes: row_shr 1, 2, 4, 8 (a lane without a source adds 0), then lane 15 of rows 0 / 2 into
 * rows 1 / 3 (row_bcast:15), then lane 31 into rows 2 and 3 (row_bcast:31) */
__device__ static inline uint32_t wscan32(uint32_t v)
{
    v += REP_DPP(0, v, 0x111, 0xf);
    v += REP_DPP(0, v, 0x112, 0xf);
    v += REP_DPP(0, v, 0x114, 0xf);
    v += REP_DPP(0, v, 0x118, 0xf);
    v += REP_DPP(0, v, 0x142, 0xa);
    v += REP_DPP(0, v, 0x143, 0xc);
    return v;
}
__device__ static inline uint32_t wsum32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wscan32(v), WAVE - 1); }
/* 64-bit sum (mod 2^64) as three sums of 22-bit pieces: each stays below 2^28 */
__device__ static inline uint64_t wsum64(uint64_t v)
{
    const uint32_t a = wsum32((uint32_t)v & 0x3FFFFFu), b = wsum32((uint32_t)(v >> 22) & 0x3FFFFFu), c = wsum32((uint32_t)(v >> 44));
    return (uint64_t)a + ((uint64_t)b << 22) + ((uint64_t)c << 44);
}
__device__ static inline uint32_t wmin32(uint32_t v)
{
    v = min(v, REP_DPP(v, v, 0x111, 0xf));
    v = min(v, REP_DPP(v, v, 0x112, 0xf));
    v = min(v, REP_DPP(v, v, 0x114, 0xf));
    v = min(v, REP_DPP(v, v, 0x118, 0xf));
    v = min(v, REP_DPP(v, v, 0x142, 0xa));
    v = min(v, REP_DPP(v, v, 0x143, 0xc));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, WAVE - 1);
}
/* the value of the lane below (lane 0: its own) */
__device__ static inline uint32_t wprev32(uint32_t v) { return REP_DPP(v, v, 0x138, 0xf); }      /* wave_shr:1 */
/* Rows that the lanes of a wavefront hold (row r in lane r) go to memory as WHOLE LINES: instruction j writes
 * 1 KiB contiguous, lane l the 16 bytes [l * 16, l * 16 + 16) of it -- a write-through store that covers part of
 * a line is a read-modify-write in HBM, eight 8-byte stores to one line are eight of them in a row. */
__device__ static inline uint4 rl128(uint4 v, int l)
{
    return make_uint4((uint32_t)__shfl((int)v.x, l, WAVE), (uint32_t)__shfl((int)v.y, l, WAVE), (uint32_t)__shfl((int)v.z, l, WAVE), (uint32_t)__shfl((int)v.w, l, WAVE));
}
/* 64-byte rows: w[0..7] of row r in lane r; row r goes to row_base(r) */
template <typename AddrOf>
__device__ static inline void rep_store_rows64(const uint64_t (&w)[8], uint32_t nrows, AddrOf row_addr)
{
    const uint32_t lane = lane_id(), q = lane & 3;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int src = j * 16 + (int)(lane >> 2);
        uint64_t lo = 0, hi = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint64_t a = rl64v(w[2 * c], src), b = rl64v(w[2 * c + 1], src);
            if ((uint32_t)c == q) { lo = a; hi = b; }
        }
        if ((uint32_t)src < nrows) st16_wt(row_addr((uint32_t)src) + q * 16, make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)));
    }
}
/* the same through 4 KiB of LDS (8 LDS instructions per lane instead of 64 cross-lane reads): lane r leaves its row at
 * scratch + 64 r, then reads the 16 bytes it is to store */
template <typename AddrOf>
__device__ static inline void rep_store_rows64_lds(uint4 *scratch, const uint64_t (&w)[8], uint32_t nrows, AddrOf row_addr)
{
    const uint32_t lane = lane_id(), q = lane & 3;
#pragma unroll
    for (int c = 0; c < 4; c++)
        scratch[lane * 4 + c] = make_uint4((uint32_t)w[2 * c], (uint32_t)(w[2 * c] >> 32), (uint32_t)w[2 * c + 1], (uint32_t)(w[2 * c + 1] >> 32));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t src = (uint32_t)j * 16 + (lane >> 2);
        const uint4 v = scratch[j * WAVE + lane];              /* = row src, quarter q */
        if (src < nrows) st16_wt(row_addr(src) + q * 16, v);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       /* (the rows are read before the next chunk overwrites them) */
}
/* 32-byte rows: (a, b) of row r in lane r */
template <typename AddrOf>
__device__ static inline void rep_store_rows32(uint4 a, uint4 b, uint32_t nrows, AddrOf row_addr)
{
    const uint32_t lane = lane_id(), h = lane & 1;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int src = j * 32 + (int)(lane >> 1);
        const uint4 va = rl128(a, src), vb = rl128(b, src);
        if ((uint32_t)src < nrows) st16_wt(row_addr((uint32_t)src) + h * 16, h ? vb : va);
    }
}

/* where the n entries of one round go (log_append_entry, dare_log.h:466-558): lane j holds the size T of
 * entry j (0 beyond n), e0 = the log's end before the round (len: it reads as empty).  The entry that does
 * not fit before len wraps to offset 0 -- leaving a stale header behind when only its payload did not fit
 * (dare_log.h:502-538); an entry that starts exactly on len finds the log "empty" and restarts idx at 1
 * (dare_log.h:158-162, 486-488: SURVEY Q13). */
struct RepPlace { int kstar, estar; uint32_t stale; uint64_t w, a, total; };
__device__ static inline RepPlace rep_place(uint64_t e0, uint64_t L, uint32_t T, uint32_t n)
{
    const bool active = lane_id() < n;
    const uint64_t incl = wscan32(T);                  /* (a round is at most 64 x 65 599 bytes) */
    RepPlace s;
    s.a = e0 + incl - T;
    const unsigned long long over = __ballot(active && s.a + T > L);
    s.kstar = -1; s.estar = -1; s.stale = 0; s.w = 0;
    if (over) {
        const int ks = __builtin_ctzll(over);
        s.kstar = ks;
        s.w = rl64u(s.a, ks);
        if (s.w == L) s.estar = ks; else if (L - s.w >= APUS_HDR) s.stale = 1;
    }
    s.total = rl64u(incl, (int)n - 1);
    return s;
}
__device__ static inline uint64_t rep_pos(const RepPlace &s, int j) { return (s.kstar < 0 || j < s.kstar) ? s.a : (j == s.kstar ? 0 : s.a - s.w); }
__device__ static inline uint64_t rep_idx(const RepPlace &s, int j, uint64_t idx0) { return (s.estar < 0 || j < s.estar) ? idx0 + (uint64_t)j : 1 + (uint64_t)(j - s.estar); }
/* the engine's log-full rule (deviation 2): a round that does not fit into the free
 * part of the ring is refused as a whole, before any store */
__device__ static inline bool rep_refuse(uint64_t e0, uint64_t L, uint64_t head, const RepPlace &s)
{
    if (e0 == L) return false;
    const uint64_t waste = s.kstar >= 0 ? L - s.w : 0;
    const uint64_t used = e0 >= head ? e0 - head : L - (head - e0);
    return e0 == head || s.total + waste > L - used;
}

/* ===================================================================================== leader */
struct RepSeqState {                    /* the sequencer's registers: the leader's append-side words */
    uint64_t end, tail, last_idx, n_end, head, prev_head, store_count;
    uint64_t c_off, c_slot;             /* the commit as it stands once everything issued has its majority */
    uint64_t sample_slot;               /* slots the servers sampled by the last tick are taken to have applied */
    uint64_t t;                         /* tickets issued                                      */
    uint64_t pass_seq;                  /* bulk passes issued                                  */
    uint64_t g_seq;                     /* ... and passes without ticket words                 */
    /* Head moves that are not verified yet (rep_seq_prune): a small queue {the slot count every sampled server must have
     * applied, the head that then holds} -- entry j in lane j of two registers (round 6: in LDS until then, like the sampled
     * apply offsets: a dozen LDS round trips per prune tick on a compute unit whose append wavefronts live in LDS) -- and the
     * head as it stands with the verified moves only: what protects the ring (rep_refuse) until the rest is verified. */
    uint64_t pv_need, pv_head;
    uint64_t ao;                        /* lane i: ctrl_data->apply_offsets[i], what the last tick sampled for server i */
    uint64_t st_pr[4];                  /* diagnostics (APUS_REP_DBG & 512): prune ticks, and their time by phase */
    bool stats;
    uint64_t head_safe;
    uint32_t pv_r, pv_n;
    uint64_t tail_round;                /* staged round whose last entry is the tail (tail worked out when somebody asks) */
    uint32_t push_mask;
    bool     can_commit, tail_known;
};
/* the offset of the last entry (log->tail): only a prune tick and the end of the run want it */
__device__ static inline void rep_seq_fix_tail(const EngDev &E, RepSeqState &S)
{
    if (S.tail_known) return;
    const uint32_t last = E.round_first[S.tail_round + 1] - 1;
    S.tail = S.end - (APUS_HDR + (uint64_t)E.req_len[last]);
    S.tail_known = true;
}

#define R_PV 8u                         /* head moves that may be unverified at a time */
struct RepSeqCtx { const RepArgs *A; lds_u64 s_m; RepBox *mybox; };
__device__ static inline bool rep_seq_wait_verified(const EngDev &E, const RepSeqCtx &X, RepSeqState &S, RepLead *LS, uint32_t keep);

__device__ static inline bool rep_quorum(const EngDev &E, uint32_t push_mask)
{
    const uint32_t size = E.group_size, size_mask = (1u << size) - 1;
    return (uint32_t)__popc((push_mask | (1u << E.leader)) & size_mask) >= size / 2 + 1;
}

/* one round through the general placement code: lane j holds its entry's size.  False: refused (log full). */
__device__ static inline bool rep_seq_round(const EngDev &E, const RepSeqCtx &X, RepSeqState &S, RepLead *LS, uint32_t T, uint32_t n, uint32_t kind,
                                            uint64_t first, uint32_t ctype, uint64_t d0, uint64_t d1)
{
    const uint64_t L = E.log_len;
    const uint32_t lane = lane_id();
    const RepPlace p = rep_place(S.end, L, T, n);
    /* the round would run into bytes that lie behind the head only by moves nobody has verified yet: verify first */
    while (S.pv_n && rep_refuse(S.end, L, S.head_safe, p)) {
        if (!rep_seq_wait_verified(E, X, S, LS, S.pv_n - 1)) return false;
    }
    if (rep_refuse(S.end, L, S.head, p)) return false;
    const int last = (int)n - 1;
    const uint64_t pos_l = rl64u(rep_pos(p, (int)lane), last);
    const uint32_t T_l = rl32u(T, last);
    const uint64_t idx_l = rep_idx(p, last, S.last_idx + 1);
    const uint64_t end_new = pos_l + T_l;
    const uint32_t hidden = end_new == L;
    uint64_t v = 0;
    switch (lane) {
    case TK_E0: v = S.end; break;
    case TK_IDX0: v = S.last_idx + 1; break;
    case TK_SLOT0: v = S.n_end; break;
    case TK_SRC: v = first; break;
    case TK_END: v = end_new; break;
    case TK_D0: v = d0; break;
    case TK_D1: v = (kind == R_SRC_CONTROL) ? d1 : (wall_clock64() & 0xFFFFFFFFull); break;
    case TK_META: v = (uint64_t)n | ((uint64_t)kind << 8) | ((uint64_t)hidden << 12) | ((uint64_t)ctype << 16) | ((uint64_t)S.push_mask << 32); break;
    default: break;
    }
    if (lane < 8) st_agent(&LS->tkw[lane][S.t % RS_CAP], rep_tk(S.t, v));
    S.t++;
    S.end = end_new; S.tail = pos_l; S.tail_known = true; S.last_idx = idx_l; S.n_end += n; S.store_count += n;
    if (!(kind == R_SRC_CONTROL && ctype == 3)) S.prev_head = 0;
    if (S.can_commit && !hidden) { S.c_off = end_new; S.c_slot = S.n_end; }
    return true;
}

/* the tickets issued so far: for the committer (LDS).  The append wavefronts poll their tickets themselves. */
__device__ static inline void rep_seq_publish(RepLead *LS, lds_u64 s_m, const RepSeqState &S, uint64_t progress)
{
    if (lane_id() == 0) {
        s_m[M_TAIL] = S.t;                             /* (whoever reads M_PROG first and M_TAIL second sees every ticket behind the progress) */
        s_m[M_PROG] = progress;
    }
}

/* a follower leaves the push set (it did not consume its doorbells / did not apply in time) */
__device__ static inline void rep_seq_drop(const EngDev &E, RepSeqState &S, RepLead *LS, uint32_t drop, uint32_t site)
{
    drop &= S.push_mask;
    if (!drop) return;
    S.push_mask &= ~drop;
    if (lane_id() == 0) {
        atomicOr((unsigned long long *)&LS->drop_mask, (unsigned long long)drop);
        for (uint32_t m = drop; m; m &= m - 1) st_agent(&LS->t_drop[__builtin_ctz(m)], S.t);
        spin_timeout(E, site);
    }
    S.can_commit = rep_quorum(E, S.push_mask);
}

/* log_pruning (dare_server.c:1996-2067) as the leader's timer tick.  The reference's timer fires between
 * polling() passes: every reachable server has applied what is committed, and the tick (a) moves the head
 * to the smallest apply offset SAMPLED BY THE PREVIOUS TICK, (b) appends <HEAD, head>, (c) samples the apply
 * offsets for the next one (rc_get_remote_apply_offsets, dare_ibv_rc.c:1970-2034).  With many rounds in
 * flight the pipeline is not drained for that: the offsets sampled are the ones the pinned schedule gives
 * (everything issued before the tick has its majority and is applied: S.c_off) and the NEXT tick first
 * verifies that every sampled server really got there (applied_by[] in the leader's mailbox, the leader's
 * own applier) before the head may move -- a server that did not is waited for (bounded), then it leaves
 * the push set and the head stays.  Without a majority nothing commits: S.c_off stands still, the sample
 * is the real state. */
/* Round 4: the tick no longer WAITS for the last tick's samples to become true (that wait drained the pipeline to
 * less than one tick's stretch of rounds seventeen times per pass over configs[1]: the append wavefronts starved, more
 * workgroups bought nothing).  The head moves at once -- its value is the pinned schedule's, nothing else depends on the
 * verification -- and the move goes into a queue {slot count the sampled servers must have applied, new head}; the
 * ring is protected by the VERIFIED head (S.head_safe) until the queue entry is retired: rep_seq_round and the plain
 * path refuse / wait on S.head_safe, a tick that finds the queue full waits for its oldest entry (bounded; a follower
 * that does not get there leaves the push set, as before). */
__device__ static inline uint64_t rep_seq_applied(const RepSeqState &S, lds_u64 s_m, RepBox *mybox)
{
    const uint32_t lane = lane_id();
    if (lane == 0) return s_m[M_N_APPLY];
    if (lane <= APUS_DEV_MAX_SERVERS && ((S.push_mask >> (lane - 1)) & 1u)) return ld_sys(&mybox->applied_by[lane - 1]);
    return ~0ull;
}
/* retires the head moves that have become true (one look at the applied counts); -> the lanes (0: the leader's own
 * applier, f + 1: follower f) that are still behind the oldest one left, 0 when nothing is left */
__device__ static inline unsigned long long rep_seq_verify(RepSeqState &S, lds_u64 s_m, RepBox *mybox, const uint64_t *pre = nullptr)
{
    if (!S.pv_n) return 0;
    /* (pre: the followers' counts as a pass asked for them a moment ago -- older counts retire fewer moves, never a wrong one) */
    const uint64_t v = pre ? (lane_id() == 0 ? (uint64_t)s_m[M_N_APPLY] : *pre) : rep_seq_applied(S, s_m, mybox);
    for (;;) {
        if (!S.pv_n) return 0;
        const int ix = (int)(S.pv_r % R_PV);
        const uint64_t need = rl64u(S.pv_need, ix);
        const unsigned long long late = __ballot(v < need);
        if (late) return late;
        S.head_safe = rl64u(S.pv_head, ix);
        S.pv_r++; S.pv_n--;
    }
}
/* waits (bounded) until at most `keep` head moves are unverified.  False: the leader's own applier did not get there. */
__device__ static inline bool rep_seq_wait_verified(const EngDev &E, const RepSeqCtx &X, RepSeqState &S, RepLead *LS, uint32_t keep)
{
    for (uint64_t spins = 0; S.pv_n > keep;) {
        const uint32_t before = S.pv_n;
        const unsigned long long late = rep_seq_verify(S, X.s_m, X.mybox);
        if (S.pv_n <= keep) break;
        if (S.pv_n != before) { spins = 0; continue; }
        if (++spins > X.A->peer_polls) {
            if (late >> 1) { rep_seq_drop(E, S, LS, (uint32_t)(late >> 1), 7101); spins = 0; continue; }
            return false;
        }
        __builtin_amdgcn_s_sleep(4);
    }
    return true;
}

/* The <HEAD> entry of a prune tick in closed form: 64 bytes at the log's end, no wrap, no exact fit, room in front of both heads
 * -- every tick but the one per lap whose entry meets len.  What rep_seq_round does for such a round, without the wave-scan
 * placement and its two kilobytes of code: the kernel is 114 KB against an instruction cache of 64 KB per two compute units,
 * and a tick runs once per 1024 rounds -- every line of code it touches is fetched from memory again (round 6: the tick's
 * <HEAD> round was 3.6 of its 4.7 us at three replicas).  False: not such a round, nothing done. */
__device__ static inline bool rep_seq_head_plain(const EngDev &E, RepSeqState &S, RepLead *LS, uint64_t new_head)
{
    const uint64_t L = E.log_len, e0 = S.end;
    if (e0 == L || e0 + APUS_HDR >= L) return false;
    const uint64_t used = e0 >= S.head ? e0 - S.head : L - (S.head - e0);
    const uint64_t used_s = e0 >= S.head_safe ? e0 - S.head_safe : L - (S.head_safe - e0);
    if (e0 == S.head || APUS_HDR > L - used) return false;
    if (S.pv_n && (e0 == S.head_safe || APUS_HDR > L - used_s)) return false;          /* (the general path waits for the verification) */
    const uint32_t lane = lane_id();
    uint64_t v = 0;
    switch (lane) {
    case TK_E0: v = e0; break;
    case TK_IDX0: v = S.last_idx + 1; break;
    case TK_SLOT0: v = S.n_end; break;
    case TK_END: v = e0 + APUS_HDR; break;
    case TK_D0: v = new_head; break;
    case TK_META: v = 1ull | ((uint64_t)R_SRC_CONTROL << 8) | (3ull << 16) | ((uint64_t)S.push_mask << 32); break;
    default: break;
    }
    if (lane < 8) st_agent(&LS->tkw[lane][S.t % RS_CAP], rep_tk(S.t, v));
    S.t++;
    S.tail = e0; S.tail_known = true; S.end = e0 + APUS_HDR; S.last_idx++; S.n_end++; S.store_count++;
    if (S.can_commit) { S.c_off = S.end; S.c_slot = S.n_end; }
    return true;
}

__device__ static inline void rep_seq_prune(const EngDev &E, const RepArgs &A, const RepSeqCtx &X, RepSeqState &S,
                                            lds_u64 s_m, uint32_t bitmask, RepBox *mybox, uint64_t progress, const uint64_t *pre = nullptr)
{
    RepLead *LS = A.LS;
    const uint64_t L = E.log_len;
    const uint32_t lane = lane_id();
    /* (0) the earlier ticks' samples: what has become true is retired; with the queue full the oldest is waited for */
    bool late = false;
    const uint64_t tq0 = S.stats ? wall_clock64() : 0;
    rep_seq_verify(S, s_m, mybox, pre);
    if (S.pv_n >= R_PV) late = !rep_seq_wait_verified(E, X, S, LS, R_PV - 1);
    const uint64_t tq1 = S.stats ? wall_clock64() : 0;
    const uint32_t size = E.group_size;
    const uint64_t c_before = S.c_off, cs_before = S.c_slot;
    if (!late) {
        /* (a) + (b) */
        uint64_t min_off = S.c_off;                                    /* the leader's own apply offset */
        if (lane < size && !((bitmask >> lane) & 1u)) S.ao = S.c_off;
        for (uint32_t i = 0; i < size; i++) {
            const uint64_t ao_i = rl64u(S.ao, (int)i);
            if (apus_is_larger(S.end, L, min_off, ao_i)) min_off = ao_i;
        }
        if (apus_end_distance(S.end, L, min_off) == 0) { rep_seq_fix_tail(E, S); min_off = S.tail; }      /* leave one entry, :2038-2041 */
        if (apus_is_larger(S.end, L, min_off, S.head) && !S.prev_head) {
            const uint64_t head_before = S.head;
            S.head = min_off;
            if (rep_seq_head_plain(E, S, LS, min_off) || rep_seq_round(E, X, S, LS, lane == 0 ? APUS_HDR : 0u, 1, R_SRC_CONTROL, 0, 3, min_off, 0)) {
                S.prev_head = 1;
                /* the move holds once every sampled server has applied what the LAST tick took it to have applied */
                const uint32_t ix = (S.pv_r + S.pv_n) % R_PV;
                if (lane == ix) { S.pv_need = S.sample_slot; S.pv_head = S.head; }
                S.pv_n++;
            } else { S.head = head_before; if (lane == 0) { set_status(E, 1u << 1); st_sys(&A.H->full, ld_sys(&A.H->full) + 1); } }
        }
    }
    const uint64_t tq2 = S.stats ? wall_clock64() : 0;
    /* (c): what the servers will have applied when the timer's pass is over = the commit before <HEAD> */
    if (lane < size && (lane == E.leader || !((bitmask >> lane) & 1u) || ((S.push_mask >> lane) & 1u))) S.ao = c_before;
    S.sample_slot = cs_before;
    rep_seq_publish(LS, s_m, S, progress);
    if (S.stats) { S.st_pr[0]++; S.st_pr[1] += tq1 - tq0; S.st_pr[2] += tq2 - tq1; S.st_pr[3] += wall_clock64() - tq2; }
}

/* the leader's first workgroup: wavefront 0 sequences, wavefront 1 commits, wavefront 2 applies */
__device__ static inline void rep_sequencer(const EngDev &E, const RepArgs &A, lds_u64 s_h, lds_u64 s_ao,
                                            lds_u64 s_m, lds_u64 s_x, uint4 *s_tr /* 4 KiB of LDS: the queue of unverified head moves */)
{
    REP_SERIAL_PRIO();
    RepHost *H = A.H;
    RepReq *RQ = A.RQ;
    RepLead *LS = A.LS;
    RepBox *mybox = E.box[E.leader];
    const uint64_t L = E.log_len;
    const uint32_t lane = lane_id();
    RepSeqState S;
    S.end = s_h[H_END]; S.tail = s_h[H_TAIL]; S.last_idx = s_h[H_LAST_IDX]; S.n_end = s_h[H_N_END]; S.head = s_h[H_HEAD];
    S.prev_head = s_h[H_PREV_HEAD]; S.store_count = s_h[H_STORE_COUNT];
    S.c_off = s_h[H_COMMIT]; S.c_slot = s_h[H_N_COMMIT]; S.sample_slot = 0; S.t = 0; S.pass_seq = 0; S.g_seq = 0; S.tail_known = true; S.tail_round = 0;
    S.push_mask = A.push_mask; S.can_commit = rep_quorum(E, S.push_mask);
    S.pv_need = 0; S.pv_head = 0; S.head_safe = S.head; S.pv_r = 0; S.pv_n = 0;
    S.st_pr[0] = S.st_pr[1] = S.st_pr[2] = S.st_pr[3] = 0; S.stats = (A.dbg & 512) != 0;
    S.ao = lane < APUS_DEV_MAX_SERVERS ? (uint64_t)s_h[H_APPLY_OFFSETS + lane] : 0ull;
    const RepSeqCtx X = {&A, s_m, mybox};
    const uint32_t bitmask = (uint32_t)s_h[H_CID_BITMASK];
    uint64_t req_head = ld_sys(&H->slots_done), cmd_head = ld_sys(&H->cmd_head);
    /* The host's next commands: up to SIXTEEN per look at the command ring, held in one register -- lane l: granule l & 3 of command
     * cq_base + (l >> 2) -- and taken from there (round 5 fetched two per look: one round trip through the mailbox's memory per RUN +
     * PRUNE pair, 1.3 us of the ~10 us the sequencer spent per pair at three replicas).  cq_n of them are all there, cq_i are carried out. */
    uint64_t cq = 0, cq_base = 0;
    uint32_t cq_n = 0, cq_i = 0;
    bool have_cmd = false;
    uint32_t cmd_op = 0; uint64_t cmd_after = 0, cmd_a = 0, cmd_b = 0;
    /* granules just loaded for commands `base` ...: how many commands in a row are complete */
    auto cq_take = [&](uint64_t w, uint64_t base) {
        if (cq_i < cq_n) return;                       /* (what is queued goes first; the words are asked for again when it is used up) */
        unsigned long long m = __ballot(rep_gran_ok(w, base + (lane >> 2)));
        m &= m >> 1; m &= m >> 2; m &= 0x1111111111111111ull;
        const unsigned long long y = ~m & 0x1111111111111111ull;
        cq = w; cq_base = base; cq_i = 0; cq_n = y ? (uint32_t)__builtin_ctzll(y) >> 2 : 16u;
    };
    /* the next command into the first register */
    auto cq_front = [&]() {
        if (have_cmd || cq_i >= cq_n) return;
        const int l = (int)(4 * cq_i);
        have_cmd = true; cq_i++;
        cmd_op = (uint32_t)rl64u(cq, l); cmd_after = rep_extend(req_head, (uint32_t)rl64u(cq, l + 1));
        cmd_a = (uint32_t)rl64u(cq, l + 2); cmd_b = (uint32_t)rl64u(cq, l + 3);
    };
    uint64_t run_next = 0, run_end = 0;
    uint64_t idle = 0, budget = 0, dropped = 0;
    uint32_t exit_code = R_EXIT_STOP;
    if (lane == 0) st_sys(&H->alive, 1);
    const uint64_t my_qbase = (lane >= 1 && lane <= APUS_DEV_MAX_SERVERS) ? A.qbase[lane - 1] : 0;    /* (lane f + 1 looks after follower f) */
    uint64_t st_pass = 0, st_staged = 0, st_flow = 0, st_busy = 0, st_prune = 0, st_flowt = 0, st_pcie = 0, st_pcie_n = 0, st_reload = 0;
    uint64_t st_ph[4] = {0, 0, 0, 0};
    const bool stats = A.dbg & 512;      /* per-pass clocks: every look at the wall clock is a scalar memory round trip */
    const uint64_t st_t0 = wall_clock64();

    uint64_t n_pf0 = 0, n_pf1 = 0, n_spf = 0, pre_rc = ~0ull;      /* the next RUN's words as the pass before asked for them (rounds from pre_rc on, pre_avail of them) */
    uint32_t n_rf0 = 0, n_rf1 = 0, n_srf = 0, n_scg = 0, n_cg0 = 0, pre_avail = 0;
    uint64_t ap_v = ~0ull;                       /* the followers' applied counts as the last staged pass asked for them (the next prune tick's first look) */
    bool ap_have = false;
    bool pk_pending = false;                     /* the next host commands, asked for by a staged pass that may end its run */
    uint64_t pk_cg = 0, pk_next = 0;
    /* the request-ring passes' pipeline registers (host-fed input) */
    uint32_t pf_v[R_WIN], pf_ww = 0;
    uint64_t pf_cg = 0, pf_stop = 0, pf_head = 0, pf_cmd = 0;
    bool pf_on = false, rq_hot = false;
#pragma unroll
    for (int wdw = 0; wdw < R_WIN; wdw++) pf_v[wdw] = 0;
    auto take_peek = [&]() { cq_take(pk_cg, pk_next); };

    /* the host command in the first register, once every request slot in front of it is taken: 0 not yet, 1 carried
     * out, 2 the run ends (STOP) */
    auto exec_cmd = [&]() -> int {
        cq_front();
        if (!have_cmd || cmd_after > req_head) return 0;
        if (cmd_op == R_OP_PRUNE && budget == 0) return 0;          /* (a tick may append a <HEAD> entry: room first) */
        have_cmd = false;
        if (cmd_op == R_OP_STOP) { cmd_head++; if (lane == 0) st_sys(&H->cmd_head, cmd_head); exit_code = R_EXIT_STOP; return 2; }
        if (cmd_op == R_OP_RUN) {
            run_next = cmd_a; run_end = cmd_a + cmd_b;
            if (run_next == run_end) { cmd_head++; if (lane == 0) st_sys(&H->cmd_head, cmd_head); rep_seq_publish(LS, s_m, S, cmd_head + req_head); }
            return 1;
        }
        cmd_head++;
        if (lane == 0) st_sys(&H->cmd_head, cmd_head);
        if (cmd_op == R_OP_PRUNE) { const uint64_t tp0 = stats ? wall_clock64() : 0; rep_seq_prune(E, A, X, S, s_m, bitmask, mybox, cmd_head + req_head, ap_have ? &ap_v : nullptr); ap_have = false; budget--; if (stats) st_prune += wall_clock64() - tp0; }
        else rep_seq_publish(LS, s_m, S, cmd_head + req_head);
        return 1;
    };

    /* Flow control: room in the ticket ring and in every pushed follower's doorbell ring.  Waits until every ring has room for
     * `need` rounds (a short while when at least one round's room is there: the rest of a big pass can wait for the next look)
     * -> budget.  A follower that leaves less than a round's room for peer_polls looks in a row leaves the push set; false: the
     * leader's own ticket ring does not drain -- its commit does not move, no majority: the run ends (R_EXIT_TIMEOUT). */
    auto flow_wait = [&](uint32_t need) -> bool {
        uint64_t spins = 0, soft = 0;
        st_flow++;
        const uint64_t tf0 = stats ? wall_clock64() : 0;
        for (;;) {
            uint64_t room = ~0ull;
            if (lane == 0) {
                const uint64_t inflight = S.t - s_m[M_T_RETIRED];
                room = inflight + R_SLACK >= RS_CAP ? 0 : RS_CAP - R_SLACK - inflight;
            } else if (lane <= APUS_DEV_MAX_SERVERS && ((S.push_mask >> (lane - 1)) & 1u)) {
                const uint64_t inflight = my_qbase + S.t - ld_sys(&mybox->seqdone_by[lane - 1]);
                room = inflight + R_SLACK >= RB_CAP ? 0 : RB_CAP - R_SLACK - inflight;
            }
            const unsigned long long tight = __ballot(room < WAVE);
            const uint32_t r0 = wmin32((uint32_t)min(room, (uint64_t)0xFFFFFFFFu));
            if (r0 >= need || (!tight && ++soft > 64)) { budget = r0; break; }
            if (tight && ++spins > A.peer_polls) {
                if (tight >> 1) { rep_seq_drop(E, S, LS, (uint32_t)(tight >> 1), 7102); spins = 0; continue; }
                exit_code = R_EXIT_TIMEOUT;                     /* the leader's own commit does not move: no majority */
                return false;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        if (stats) st_flowt += wall_clock64() - tf0;
        return true;
    };

    for (;;) {
        st_pass++;
        /* ---- a host command that is already here runs at once: no look at the rings in front of it (round 3 paid a
         *      flow-control round trip and a PCIe round trip per command: a third of the sequencer's time on configs[1]) ---- */
        if (run_next == run_end) {
            const int x = exec_cmd();
            if (x == 2) break;
            if (x == 1) { idle = 0; continue; }
        }
        /* ---- flow control: room in the ticket ring and in every pushed follower's doorbell ring ---- */
        if ((budget < 4 * WAVE && run_next == run_end) || budget < WAVE) {
            if (!flow_wait(WAVE)) { if (lane == 0) spin_timeout(E, 7103); break; }
        }
        if (run_next < run_end) {
            /* ---- staged (device-resident) rounds.  Round 6: the cost of a pass does not depend on the rounds it takes.  Rounds
             *      4 and 5 gave every round of a pass a lane (256 per pass: four prefix loads per lane, the pipeline registers to hide
             *      them, 2.1-2.5 us per pass = 14.8 ns per round with the prune ticks -- the kernel's ceiling at one and three
             *      replicas, and the reason why two launches of one binary differed by 10 %: the sequencer's pass time sat right at
             *      the append wavefronts' capacity, and whoever shared its SIMD decided which of the two bound).  Now a pass takes up
             *      to GP_MAX rounds -- what is left of the run, what the rings have room for -- and needs of them only (a) the
             *      byte / request prefix at 64 CUTS of the stretch (lane l: the first avail (l + 1) / 64 rounds; the conditions
             *      under which rounds are "plain" -- no wrap, no exact fit, room in the ring -- grow with the round, so the lanes
             *      that hold form a prefix: __ballot, count trailing ones = the longest plain stretch at 1/64 granularity) and (b)
             *      two words per ticket, 64 tickets per store instruction, worked out in closed form when every round is a full
             *      one.  ONE pass record stands for all of them (TK_BULK); the append wavefronts place themselves (rep_append_wave).
             *      What is left in front of a wrap -- fewer than 64 rounds -- and the round AT the wrap go one lane per round /
             *      through the general placement, as before.  One memory round trip per pass, whatever its size. ---- */
            while (run_next < run_end) {
                const uint64_t rc = run_next;
                const uint64_t st_p0 = stats ? wall_clock64() : 0;
                st_staged++;
                const uint32_t want = (uint32_t)min((uint64_t)GP_MAX, run_end - rc);
                if (stats) st_ph[0]++;
                if (budget < want && budget < GP_GOAL) {
                    /* (room for a good part of the pass: tickets are handed out GP_GOAL at a time while >= RB_CAP - GP_GOAL rounds are queued) */
                    if (!flow_wait(min(want, (uint32_t)GP_GOAL))) break;
                }
                const uint32_t avail = (uint32_t)min((uint64_t)want, budget);
                if (!avail) break;
                const uint64_t tpa = stats ? wall_clock64() : 0;
                /* ---- everything the pass needs from memory: one round trip -- or none: the pass that ended the RUN before this one
                 *      asked for these words already when the command ring showed this RUN behind it (the round trip ran under the
                 *      prune tick in between: 1.8 of the pass's 3.6 us at three replicas) ---- */
                const uint32_t j0 = lane < avail ? lane : 0u;                            /* lane j: round rc + j (the first 64) */
                const uint32_t cut = (uint32_t)(((uint64_t)avail * (lane + 1)) >> 6);     /* lane l: the first `cut` rounds (lane 63: all) */
                uint64_t pf0, pf1, spf;
                uint32_t rf0, rf1, srf, scg, cg0;
                if (pre_rc == rc && pre_avail == avail) {
                    pf0 = n_pf0; pf1 = n_pf1; spf = n_spf; rf0 = n_rf0; rf1 = n_rf1; srf = n_srf; scg = n_scg; cg0 = n_cg0;
                } else {
                    pf0 = E.round_prefix[rc + j0]; pf1 = E.round_prefix[rc + j0 + 1];
                    rf0 = E.round_first[rc + j0]; rf1 = E.round_first[rc + j0 + 1];
                    spf = E.round_prefix[rc + cut];
                    srf = E.round_first[rc + cut];
                    scg = E.round_change[rc + (cut ? cut - 1 : 0u)]; cg0 = E.round_change[rc];      /* (one size all the way?) */
                }
                pre_rc = ~0ull;
                /* the next host commands, when this pass may end the run and none is queued */
                if (!have_cmd && cq_i >= cq_n && run_end - rc <= avail) {
                    pk_pending = true; pk_next = cmd_head + 1;            /* (cmd_head is the RUN in progress) */
                    pk_cg = ld_sys(&RQ->cmd[(pk_next + (lane >> 2)) % RC_CAP].g[lane & 3]);
                }
                /* ... and what the prune tick behind the run will want to know first: how far the followers have applied */
                ap_v = (lane >= 1 && lane <= APUS_DEV_MAX_SERVERS && ((S.push_mask >> (lane - 1)) & 1u)) ? ld_sys(&mybox->applied_by[lane - 1]) : ~0ull;
                ap_have = true;
                const uint64_t stamp = wall_clock64() & 0xFFFFFFFFull;
                const uint64_t bpf = rl64u(pf0, 0);
                const uint32_t brf = rl32u(rf0, 0);
                const uint64_t used = S.end >= S.head_safe ? S.end - S.head_safe : L - (S.head_safe - S.end);      /* (the VERIFIED head: rep_seq_prune) */
                uint32_t taken = 0;
                const uint64_t tpb = stats ? (wall_clock64() + 0 * (uint64_t)rl32u(srf, 0)) : 0;      /* (behind the loads' arrival) */
                /* ---- the longest plain stretch as ONE pass: a record + two words per ticket (TK_BULK) ---- */
                if (avail >= WAVE && !(A.dbg & 32) && rc + avail < (1ull << 32)) {
                    const uint64_t tot_c = spf - bpf;
                    const bool okc = S.end != L && S.end + tot_c < L && S.end != S.head_safe && tot_c + APUS_HDR <= L - used;
                    const unsigned long long bm = __ballot(okc);
                    const uint32_t nb = (~bm) ? (uint32_t)__builtin_ctzll(~bm) : WAVE;
                    const uint32_t N = nb ? rl32u(cut, (int)nb - 1) : 0u;
                    if (N >= WAVE) {
                        const uint64_t tot = rl64u(spf, (int)nb - 1) - bpf;
                        const uint32_t ntot = rl32u(srf, (int)nb - 1) - brf;
                        /* every round of the same size: the first request of round rc + j is brf + n_u j -- nobody needs round_first[] */
                        const uint32_t n_u = rl32u(rf1 - rf0, 0);
                        const bool all64 = rl32u(scg, (int)nb - 1) == rl32u(cg0, 0);
                        const bool wordless = N >= GP_MIN && !(A.dbg & 128);
                        const uint64_t pn = wordless ? S.g_seq++ : S.pass_seq++;
                        uint64_t v = 0;
                        switch (lane) {
                        case PR_T0: v = S.t; break;
                        case PR_RC0: v = rc | ((uint64_t)n_u << 32); break;
                        case PR_END0: v = S.end; break;
                        case PR_IDX0: v = S.last_idx + 1; break;
                        case PR_SLOT0: v = S.n_end; break;
                        case PR_BPF: v = bpf; break;
                        case PR_BRF_N: v = (uint64_t)brf | ((uint64_t)N << 32) | (all64 ? PR_UNIFORM : 0ull); break;
                        case PR_PUSH_STAMP: v = (uint64_t)(uint32_t)stamp | ((uint64_t)S.push_mask << 32); break;
                        default: break;
                        }
                        if (lane < 8) st_agent(wordless ? &LS->grec[pn % GR_CAP][lane] : &LS->prec[pn % PR_CAP][lane], rep_tk(pn, v));
                        if (!wordless) {
                        /* the tickets' two words: ticket S.t + j = round rc + j.  (the tag of ticket t is t % 65535 + 1: worked out
                         * from the pass's first ticket with 32-bit arithmetic -- a 64-bit modulo per word was a third of the old pass) */
                        const uint32_t tbase = (uint32_t)(S.t % 65535ull), t_lo = (uint32_t)S.t;
                        const uint64_t metab = TK_BULK | (pn & 0xFFFull);
                        auto put = [&](uint32_t j, uint32_t a, uint32_t b) {
                            const uint32_t x = tbase + j;                                /* (j < GP_MAX: below 2 x 65535) */
                            const uint64_t tag = (uint64_t)((x >= 65535u ? x - 65535u : x) + 1u) << 48;
                            const uint32_t ix = (t_lo + j) & (RS_CAP - 1u);
                            st_agent(&LS->tkw[TK_SRC][ix], tag | (uint64_t)a | ((uint64_t)(b - a) << 32));
                            st_agent(&LS->tkw[TK_META][ix], tag | metab | ((rc + j) << 16));
                        };
                        put(lane, rf0, rf1);                                             /* (N >= 64: every lane) */
                        for (uint32_t c0 = 1; c0 * WAVE < N; c0 += GP_GRP) {
                            uint32_t a[GP_GRP], b[GP_GRP];
                            if (all64) {
#pragma unroll
                                for (int g = 0; g < GP_GRP; g++) { a[g] = brf + ((c0 + (uint32_t)g) * WAVE + lane) * n_u; b[g] = a[g] + n_u; }
                            } else {
#pragma unroll
                                for (int g = 0; g < GP_GRP; g++) {
                                    const uint32_t j = (c0 + (uint32_t)g) * WAVE + lane;
                                    const uint64_t r = rc + (j < N ? j : 0u);
                                    a[g] = E.round_first[r]; b[g] = E.round_first[r + 1];
                                }
                            }
#pragma unroll
                            for (int g = 0; g < GP_GRP; g++) {
                                const uint32_t j = (c0 + (uint32_t)g) * WAVE + lane;
                                if (j < N) put(j, a[g], b[g]);
                            }
                        }
                        }
                        S.end += tot; S.last_idx += ntot; S.n_end += ntot; S.store_count += ntot; S.prev_head = 0;
                        S.tail_known = false; S.tail_round = rc + N - 1;
                        if (S.can_commit) { S.c_off = S.end; S.c_slot = S.n_end; }
                        S.t += N; taken = N;
                    }
                }
                if (!taken) {
                    /* ---- fewer than 64 rounds, or the stretch in front of / at a wrap: one lane per round of the first 64 ---- */
                    const uint32_t nch = min((uint32_t)WAVE, avail);
                    const bool on = lane < nch;
                    /* lanes that can go without the general code: the log does not read as empty, no entry of rounds
                     * 0..j crosses or touches len, everything fits into the free part of the ring */
                    const bool plain = on && S.end != L && S.end + (pf1 - bpf) < L && S.end != S.head_safe && (pf1 - bpf) + APUS_HDR <= L - used;
                    const unsigned long long pm = __ballot(plain);
                    const uint32_t np = (~pm) ? (uint32_t)__builtin_ctzll(~pm) : WAVE;      /* prefix of plain rounds */
                    if (np) {
                        if (lane < np) {
                            const uint64_t tk = S.t + lane, ix = tk % RS_CAP;
                            st_agent(&LS->tkw[TK_E0][ix], rep_tk(tk, S.end + (pf0 - bpf)));
                            st_agent(&LS->tkw[TK_IDX0][ix], rep_tk(tk, S.last_idx + 1 + (rf0 - brf)));
                            st_agent(&LS->tkw[TK_SLOT0][ix], rep_tk(tk, S.n_end + (rf0 - brf)));
                            st_agent(&LS->tkw[TK_SRC][ix], rep_tk(tk, (uint64_t)rf0));
                            st_agent(&LS->tkw[TK_END][ix], rep_tk(tk, S.end + (pf1 - bpf)));
                            st_agent(&LS->tkw[TK_D0][ix], rep_tk(tk, 0ull));
                            st_agent(&LS->tkw[TK_D1][ix], rep_tk(tk, stamp));
                            st_agent(&LS->tkw[TK_META][ix], rep_tk(tk, (uint64_t)(rf1 - rf0) | ((uint64_t)R_SRC_STAGED << 8) | ((uint64_t)S.push_mask << 32)));
                        }
                        const uint64_t tot = rl64u(pf1, (int)np - 1) - bpf;
                        const uint32_t ntot = rl32u(rf1, (int)np - 1) - brf;
                        S.end += tot; S.last_idx += ntot; S.n_end += ntot; S.store_count += ntot; S.prev_head = 0;
                        S.tail_known = false; S.tail_round = rc + np - 1;
                        if (S.can_commit) { S.c_off = S.end; S.c_slot = S.n_end; }
                        S.t += np; taken = np;
                    } else {
                        /* the round at the head of the pass needs the general code (wrap, exact fit, nearly full) */
                        const uint32_t n = rl32u(rf1 - rf0, 0);
                        const uint32_t T = lane < n ? APUS_HDR + (uint32_t)E.req_len[brf + lane] : 0u;
                        if (!rep_seq_round(E, X, S, LS, T, n, R_SRC_STAGED, brf, 0, 0, 0)) {
                            if (lane == 0) { set_status(E, 1u << 1); st_sys(&H->full, ld_sys(&H->full) + 1); }
                        }
                        taken = 1;
                    }
                }
                run_next += taken; budget -= min((uint64_t)taken, budget);
                if (run_next == run_end) { cmd_head++; if (lane == 0) st_sys(&H->cmd_head, cmd_head); }
                rep_seq_publish(LS, s_m, S, cmd_head + req_head);
                if (pk_pending) { take_peek(); pk_pending = false; }
                {
                    /* the RUN behind this one (behind the prune tick, as a rule): its words are asked for now.  Straight-line code: a
                     * load inside a branch is waited for where the branch ends, whoever needs it (the pass would pay the round trip it
                     * is meant to hide); a pass that has no RUN to look forward to asks for its own first words again, and drops them */
                    const bool q0 = cq_i < cq_n, q1 = cq_i + 1 < cq_n;
                    const int l0 = (int)min(4u * cq_i, 56u), l1 = l0 + 4;
                    const uint32_t op0 = (uint32_t)rl64u(cq, l0), op1 = (uint32_t)rl64u(cq, l1);
                    const uint64_t a0 = (uint32_t)rl64u(cq, l0 + 2), b0 = (uint32_t)rl64u(cq, l0 + 3), a1 = (uint32_t)rl64u(cq, l1 + 2), b1 = (uint32_t)rl64u(cq, l1 + 3);
                    const bool use0 = q0 && op0 == R_OP_RUN && b0 != 0, use1 = !use0 && q0 && op0 == R_OP_PRUNE && q1 && op1 == R_OP_RUN && b1 != 0;
                    const bool fwd = run_next == run_end && !have_cmd && (use0 || use1);
                    const uint64_t na = fwd ? (use0 ? a0 : a1) : rc;
                    const uint32_t nav = fwd ? (uint32_t)min((uint64_t)GP_MAX, use0 ? b0 : b1) : 1u;
                    const uint32_t nj0 = lane < nav ? lane : 0u;
                    const uint32_t ncut = (uint32_t)(((uint64_t)nav * (lane + 1)) >> 6);
                    n_pf0 = E.round_prefix[na + nj0]; n_pf1 = E.round_prefix[na + nj0 + 1];
                    n_rf0 = E.round_first[na + nj0]; n_rf1 = E.round_first[na + nj0 + 1];
                    n_spf = E.round_prefix[na + ncut];
                    n_srf = E.round_first[na + ncut];
                    n_scg = E.round_change[na + (ncut ? ncut - 1 : 0u)]; n_cg0 = E.round_change[na];
                    pre_rc = fwd ? na : ~0ull; pre_avail = nav;
                }
                if (stats) { const uint64_t tpc = wall_clock64(); st_busy += tpc - st_p0; st_ph[1] += tpa - st_p0; st_ph[2] += tpb - tpa; st_ph[3] += tpc - tpb; }
            }
            if (exit_code == R_EXIT_TIMEOUT) { if (lane == 0) spin_timeout(E, 7103); break; }
            idle = 0;
            continue;
        }
        /* ---- one round trip: the next host command, the stop word and R_WIN windows of the request ring -- unless the pass
         *      before asked for exactly these words already (a pass of full windows knows where the next one starts: its loads
         *      are in flight while its record and tickets are stored; round 5) ---- */
        const uint64_t tq0 = stats ? wall_clock64() : 0;
        st_pcie_n++;
        uint32_t v[R_WIN], ww = 0;
        uint64_t cg = 0, stopw = 0;
        if (pf_on && pf_head == req_head && pf_cmd == cmd_head + (uint64_t)have_cmd) {
#pragma unroll
            for (int wdw = 0; wdw < R_WIN; wdw++) v[wdw] = pf_v[wdw];
            cg = pf_cg; stopw = pf_stop; ww = pf_ww;
        } else {
            /* the window words of the next 64 windows, one lane each (when the ring's head stands on a window boundary) */
            if (!(req_head & (WAVE - 1))) ww = ld_sys32(&RQ->ready_win[(req_head / WAVE + lane) % (RQ_CAP / WAVE)]);
            /* (a ring that had no full window last time is looked at one window at a time: a lone request does not wait for
             * R_WIN windows of words nobody has written -- a pass over 32 empty windows is ~7 us) */
            const int nw = rq_hot ? R_WIN : 1;
#pragma unroll
            for (int wdw = 0; wdw < R_WIN; wdw++) { if (wdw < nw) v[wdw] = ld_sys32(&RQ->ready_len[(req_head + (uint64_t)wdw * WAVE + lane) % RQ_CAP]); else v[wdw] = 0u; }   /* (0: no slot's tag) */
            cg = ld_sys(&RQ->cmd[(cmd_head + (uint64_t)have_cmd + (lane >> 2)) % RC_CAP].g[lane & 3]);      /* (the commands behind the one in the first register) */
            stopw = ld_sys(&RQ->stop);                   /* (with the rings: a look at host memory would be the one PCIe round trip of the pass) */
            pf_cmd = cmd_head + (uint64_t)have_cmd;
        }
        pf_on = false;
        cq_take(cg, pf_cmd);
        if (stats) st_pcie += wall_clock64() - tq0;
        {
            const int x = exec_cmd();
            if (x == 2) break;
            if (x == 1) { idle = 0; continue; }
        }
        /* ---- the pinned request ring: whatever is published, in rounds of <= 64 (one polling() pass takes
         *      the whole tailq, dare_ibv_ud.c:780-790) ---- */
        const uint64_t limit = have_cmd ? cmd_after : ~0ull;
        bool any = false;
        /* ---- several FULL windows of equally long requests (producers that keep the ring filled): one pass record for all of
         *      them, two words per round -- the rounds are the same rounds of 64 the loop below would make one by one, at ~1 us
         *      of the sequencer each (round 3 / 4: the host-fed ceiling, 46 M entries/s whatever the number of producers) ---- */
        if (!(A.dbg & 32) && budget >= 2) {
            /* lane l: window l from the head on is all there, every request as long as the first window's */
            const uint32_t len0 = rl32u(ww & 0xFFFFu, 0);
            const uint64_t slot0 = req_head + (uint64_t)lane * WAVE;
            const bool okw = !(req_head & (WAVE - 1)) && slot0 + WAVE <= limit && (ww >> 16) == rep_slot_tag(slot0) && (ww & 0xFFFFu) == len0;
            const unsigned long long bw = __ballot(okw);
            uint32_t Wn = (~bw) ? (uint32_t)__builtin_ctzll(~bw) : WAVE;
            if ((uint64_t)Wn > budget) Wn = (uint32_t)budget;
            const uint64_t T = APUS_HDR + (uint64_t)len0;
            const uint64_t used = S.end >= S.head_safe ? S.end - S.head_safe : L - (S.head_safe - S.end);
            if (Wn >= 2 && S.end != L && S.end != S.head_safe) {
                /* as many of them as go in front of len and into the free part of the ring (the rest, and the round at the wrap, later) */
                const uint64_t per = (uint64_t)WAVE * T;
                const uint64_t r1 = (L - 1 - S.end) / per, r2 = L - used >= APUS_HDR ? (L - used - APUS_HDR) / per : 0;
                if (r1 < Wn) Wn = (uint32_t)r1;
                if (r2 < Wn) Wn = (uint32_t)r2;
            }
            const uint64_t tot = (uint64_t)Wn * WAVE * T;
            if (Wn >= 2 && S.end != L && S.end + tot < L && S.end != S.head_safe && tot + APUS_HDR <= L - used) {
                {   /* what the next pass will look at: asked for now, looked at then (words published later are seen a pass later) */
                    const uint64_t nh = req_head + (uint64_t)Wn * WAVE;
                    pf_ww = ld_sys32(&RQ->ready_win[(nh / WAVE + lane) % (RQ_CAP / WAVE)]);
#pragma unroll
                    for (int wdw = 0; wdw < R_WIN; wdw++) pf_v[wdw] = ld_sys32(&RQ->ready_len[(nh + (uint64_t)wdw * WAVE + lane) % RQ_CAP]);
                    pf_cmd = cmd_head + (uint64_t)have_cmd;
                    pf_cg = ld_sys(&RQ->cmd[(pf_cmd + (lane >> 2)) % RC_CAP].g[lane & 3]);
                    pf_stop = ld_sys(&RQ->stop);
                    pf_on = true; pf_head = nh;
                    rq_hot = true;
                }
                const uint64_t pn = S.pass_seq++;
                const uint64_t stamp = wall_clock64() & 0xFFFFFFFFull;
                uint64_t pv = 0;
                switch (lane) {
                case PR_T0: pv = S.t; break;
                case PR_RC0: pv = 0; break;
                case PR_END0: pv = S.end; break;
                case PR_IDX0: pv = S.last_idx + 1; break;
                case PR_SLOT0: pv = S.n_end; break;
                case PR_BPF: pv = T; break;
                case PR_BRF_N: pv = (req_head & 0xFFFFFFFFull) | ((uint64_t)Wn << 32); break;
                case PR_PUSH_STAMP: pv = stamp | ((uint64_t)S.push_mask << 32); break;
                default: break;
                }
                if (lane < 8) st_agent(&LS->prec[pn % PR_CAP][lane], rep_tk(pn, pv));
                if (lane < Wn) {
                    const uint64_t tk = S.t + lane;
                    st_agent(&LS->tkw[TK_SRC][tk % RS_CAP], rep_tk(tk, ((req_head + (uint64_t)lane * WAVE) & 0xFFFFFFFFull) | ((uint64_t)WAVE << 32)));
                    st_agent(&LS->tkw[TK_META][tk % RS_CAP], rep_tk(tk, TK_BULK | TK_BULK_PIN | (pn & 0xFFFull) | ((uint64_t)lane << 16)));
                }
                const uint64_t ne = (uint64_t)Wn * WAVE;
                S.end += tot; S.tail = S.end - T; S.tail_known = true; S.last_idx += ne; S.n_end += ne; S.store_count += ne; S.prev_head = 0;
                if (S.can_commit) { S.c_off = S.end; S.c_slot = S.n_end; }
                S.t += Wn; req_head += ne; budget -= Wn;
                rep_seq_publish(LS, s_m, S, cmd_head + req_head);
                idle = 0;
                continue;
            }
        }
#pragma unroll
        for (int wdw = 0; wdw < R_WIN; wdw++) {
            if (budget == 0) break;
            const uint64_t slot = req_head + lane;
            /* (while full windows are coming in, a round that starts inside a window ends on its boundary: the windows behind it
             *  then go 64 at a look by their words -- the ring's head only stands inside a window after a submit whose size is not a
             *  multiple of 64) */
            const bool ok = slot < limit && (v[wdw] >> 16) == rep_slot_tag(slot) && (!rq_hot || !(req_head & (WAVE - 1)) || (slot & ~(uint64_t)(WAVE - 1)) == (req_head & ~(uint64_t)(WAVE - 1)));
            const unsigned long long bal = __ballot(ok);
            const uint32_t n = (~bal) ? (uint32_t)__builtin_ctzll(~bal) : WAVE;
            if (n == 0) break;
            any = true;
            const uint32_t T = lane < n ? APUS_HDR + (v[wdw] & 0xFFFFu) : 0u;
            /* (TK_D0 of a round from the request ring: its entries' one size, 0 when they differ -- REP_SPEC_PAY) */
            const uint32_t T0s = rl32u(T, 0);
            const uint64_t d0u = (REP_SPEC_PAY && !__ballot(lane < n && T != T0s)) ? (uint64_t)T0s : 0ull;
            if (!rep_seq_round(E, X, S, LS, T, n, R_SRC_PINNED, req_head, 0, d0u, 0)) {
                /* refused: the requests are dropped (get_tailq_message frees the node anyway), the host is told */
                if (lane == 0) { set_status(E, 1u << 1); st_sys(&H->full, ld_sys(&H->full) + 1); }
                dropped += n;                                                        /* slots consumed without a ticket */
            } else budget--;
            req_head += n;
            rq_hot = n == WAVE;
            if (n < WAVE) break;
        }
        if (any) { if (lane == 0) s_m[M_DROPPED] = dropped; rep_seq_publish(LS, s_m, S, cmd_head + req_head); idle = 0; continue; }
        rq_hot = false;
        /* ---- nothing to do ---- */
        if (stopw) { exit_code = R_EXIT_STOP; break; }
        if (++idle > A.idle_polls) { exit_code = R_EXIT_IDLE; break; }
        rep_nap(idle < 64);
    }
    rep_seq_publish(LS, s_m, S, cmd_head + req_head);
    rep_seq_fix_tail(E, S);
    if (lane < APUS_DEV_MAX_SERVERS) E.rep[E.leader].hdr[H_APPLY_OFFSETS + lane] = S.ao;
    if (lane == 0) {
        st_agent(&LS->seq_final, S.t);
        s_m[M_FINAL] = S.t;
        LS->stat[0][0] = st_pass; LS->stat[0][1] = st_staged; LS->stat[0][2] = S.t; LS->stat[0][3] = wall_clock64() - st_t0; LS->stat[0][4] = st_flow; LS->stat[0][5] = st_busy; LS->stat[0][6] = st_prune;
        LS->stat[4][0] = st_flowt; LS->stat[4][1] = st_pcie; LS->stat[4][2] = st_pcie_n; LS->stat[4][3] = st_reload;
        LS->stat[5][0] = st_ph[0]; LS->stat[5][1] = st_ph[1]; LS->stat[5][2] = st_ph[2]; LS->stat[5][3] = st_ph[3];
        LS->stat[5][4] = S.st_pr[0]; LS->stat[5][5] = S.st_pr[1]; LS->stat[5][6] = S.st_pr[2]; LS->stat[5][7] = S.st_pr[3];
        /* the leader's append-side words (log_append_entry's bookkeeping, persist_new_entries' own part) */
        uint64_t *mh = E.rep[E.leader].hdr;
        mh[H_END] = S.end; mh[H_TAIL] = S.tail; mh[H_LAST_IDX] = S.last_idx; mh[H_N_END] = S.n_end; mh[H_HEAD] = S.head;
        mh[H_PREV_HEAD] = S.prev_head; mh[H_STORE_COUNT] = S.store_count; mh[H_OLD_END] = S.end; mh[H_N_PERSIST] = S.n_end;

        s_x[0] = exit_code; s_x[1] = S.t; s_x[2] = S.push_mask; s_x[3] = S.end; s_x[4] = S.n_end;
    }
}

/* ---- the ACK scan (update_remote_logs, dare_ibv_rc.c:1725-1758) over a window of 64 x W x 8 entries -------
 * Lane l, word q looks at the eight consecutive entry slots base + (q * 64 + l) * 8 ...: one 8-byte load per
 * follower from its ACK byte map in the leader's memory.  Per entry: replies = #{followers whose byte carries
 * this lap's tag} + 1 (the leader itself), committed iff replies >= size / 2 + 1 (:1738) -- counted for all
 * eight entries at once in the bytes of a 64-bit word; per lane the count of trailing ones; the lanes whose
 * eight entries are all there by __ballot, count-trailing-ones again: the scan stops at the first entry that
 * lacks its majority, exactly like the reference's loop. */
template <int F, int W> struct RepAckWin { uint64_t a[F][W]; uint64_t base; };

template <int F, int W>
__device__ static inline void rep_ack_load(RepAckWin<F, W> &win, const uint8_t *ackb, uint64_t cap, uint32_t dir_mask, uint32_t members, uint64_t cs)
{
    const uint32_t lane = lane_id();
    win.base = cs & ~7ull;
    uint32_t m = members;
#pragma unroll
    for (int j = 0; j < F; j++) {
        const bool has = m != 0;
        const uint32_t f = has ? (uint32_t)__builtin_ctz(m) : 0;
        m &= m - 1;
#pragma unroll
        for (int q = 0; q < W; q++) {
            const uint32_t di = (uint32_t)(win.base + (uint64_t)(q * WAVE + lane) * 8) & dir_mask;
            win.a[j][q] = has ? ld_sys((const uint64_t *)(ackb + (uint64_t)f * cap + di)) : 0ull;
        }
    }
}

/* -> the slot up to which every entry from cs on has its majority (<= vis) */
template <int F, int W>
__device__ static inline uint64_t rep_ack_eval(const RepAckWin<F, W> &win, uint32_t dir_mask, uint32_t members, uint32_t quorum, uint64_t cs, uint64_t vis)
{
    const uint32_t lane = lane_id();
    const uint64_t ONES = 0x0101010101010101ull, HI = 0x8080808080808080ull, LO7 = 0x7F7F7F7F7F7F7F7Full;
    const uint64_t K = (uint64_t)(0x80u - min(quorum - 1, 0x80u)) * ONES;     /* byte + K reaches bit 7 iff byte >= quorum - 1 */
    uint64_t total = 0;
    bool stop = false;
#pragma unroll
    for (int q = 0; q < W; q++) {
        const uint64_t g0 = win.base + (uint64_t)(q * WAVE + lane) * 8;
        const uint64_t want = (uint64_t)rep_ack_tag(g0, dir_mask) * ONES;
        uint64_t cnt = 0;
        uint32_t m = members;
#pragma unroll
        for (int j = 0; j < F; j++) {
            if (m) {
                const uint64_t t = win.a[j][q] ^ want;                        /* a zero byte = this lap's tag = ACK */
                const uint64_t nz = (((t & LO7) + LO7) | t) & HI;
                cnt += ((~nz) & HI) >> 7;
            }
            m &= m - 1;
        }
        uint64_t okb = (cnt + K) & HI;                                        /* popcount(acks | self) >= size/2 + 1, per byte */
        const int nb = cs > g0 ? (int)min(cs - g0, (uint64_t)8) : 0;          /* entries below cs are committed already */
        const int nv = vis > g0 ? (int)min(vis - g0, (uint64_t)8) : 0;        /* entries from vis on do not exist yet     */
        okb |= byte_mask64(0, nb) & HI;
        okb &= byte_mask64(0, nv);
        const uint64_t bad = ~okb & HI;
        const uint32_t pre = bad ? (uint32_t)__builtin_ctzll(bad) >> 3 : 8u;  /* trailing ones of this lane's eight */
        const unsigned long long bal = __ballot(pre == 8);
        const uint32_t nfull = (~bal) ? (uint32_t)__builtin_ctzll(~bal) : WAVE;
        if (!stop) {
            total += (uint64_t)nfull * 8;
            if (nfull < WAVE) { total += rl32u(pre, (int)nfull); stop = true; }
        }
    }
    const uint64_t upto = win.base + total;
    return upto <= cs ? cs : (upto < vis ? upto : vis);
}

struct RepCommitState {
    uint64_t t_done;                    /* tickets whose bytes are in every pushed ring */
    uint64_t vis, vis_off, n_end_seen, cs, slots_done;
    uint64_t pre_end;                   /* entries below this slot were appended before this run: no ticket stands for them */
    uint32_t push_live;
    bool progress;
};

/* One pass of the committer = ONE memory round trip: the done granules of the next R_SUB x 64 tickets (which rounds'
 * bytes are in every pushed ring, in order: what is visible of the log) and the line of the followers' cumulative
 * ACKs.  replies for an entry = #{followers that hold everything up to it} + 1 (the leader itself); committed iff
 * replies >= size / 2 + 1 (dare_ibv_rc.c:1738) -- with in-order ACKs that is: everything below the (quorum - 1)-th
 * largest of the followers' counts; the scan stops at the first entry that lacks its majority (:1741) = at that count.
 * Granules may be looked at before they are there: one that is not this round's does not count. */
__device__ static inline void rep_commit_pass(const EngDev &E, RepLead *LS, RepCommitState &C, uint64_t tail,
                                              const RepBox *mybox, uint32_t members, uint32_t quorum, uint64_t my_tag, uint64_t term_slot0)
{
    const uint32_t lane = lane_id();
    const uint64_t base = C.t_done;
    uint64_t g0[R_SUB], g1[R_SUB], g2[R_SUB];
    /* (as many chunks of 64 as tickets are out: a lone round is one chunk's loads, not R_SUB chunks' -- the issue time of 24
     *  loads nobody needs sat in every hop of a lone round's latency) */
    const uint32_t nsub = (uint32_t)min((uint64_t)R_SUB, tail > base ? (tail - base + WAVE - 1) / WAVE : 1ull);
#pragma unroll
    for (int s = 0; s < R_SUB; s++) {
        g0[s] = 0; g1[s] = 0; g2[s] = 0;
        if ((uint32_t)s < nsub) {
            const uint64_t ix = (base + (uint64_t)s * WAVE + lane) % RS_CAP;
            g0[s] = ld_agent(&LS->dn[DN_META][ix]); g1[s] = ld_agent(&LS->dn[DN_SLOT_END][ix]); g2[s] = ld_agent(&LS->dn[DN_END][ix]);
        }
    }
    const uint64_t pbw = lane < 16 ? ld_sys(&mybox->persisted_by[lane]) : 0ull;
#if REP_FAST_ACK
    const uint64_t pbf = lane < 16 ? ld_sys(&mybox->persisted_fast_by[lane]) : 0ull;
#endif
#pragma unroll
    for (int s = 0; s < R_SUB; s++) {
        if (C.t_done != base + (uint64_t)s * WAVE || (uint32_t)s >= nsub) break;           /* (chunk s is looked at only when every chunk before it is in whole) */
        const uint64_t k = base + (uint64_t)s * WAVE + lane;
        const bool ret = k < tail && rep_gran_ok(g0[s], k) && rep_gran_ok(g1[s], k) && rep_gran_ok(g2[s], k);
        const unsigned long long balr = __ballot(ret);
        const uint32_t pr = (~balr) ? (uint32_t)__builtin_ctzll(~balr) : WAVE;
        if (!pr) break;
        /* ---- rounds whose bytes are in every pushed ring, in order ---- */
        const uint32_t meta = (uint32_t)g0[s];
        const uint32_t n = meta & 0xFF;
        const uint32_t pinned = (lane < pr && ((meta >> 8) & 0xF) == R_SRC_PINNED) ? n : 0;
        if (__ballot(pinned != 0)) C.slots_done += wsum32(pinned);
        const uint32_t meta_l = rl32u(meta, (int)pr - 1);
        const uint32_t se_l = rl32u((uint32_t)g1[s], (int)pr - 1);
        const uint32_t end_l = rl32u((uint32_t)g2[s], (int)pr - 1);
        const uint64_t slot_end = rep_extend(C.n_end_seen, se_l);
        C.n_end_seen = slot_end;
        C.t_done += pr;
        if ((uint64_t)end_l == E.log_len) {
            /* the round sits exactly on len: the log reads as empty, nothing of it is visible until the next one is in */
            C.vis = slot_end - (meta_l & 0xFF);
            C.vis_off = ld_agent(&LS->tkw[TK_E0][(C.t_done - 1) % RS_CAP]) & TK_VAL;
        } else { C.vis = slot_end; C.vis_off = end_l; }
        C.push_live = meta_l >> 16;
        C.progress = true;
        if (pr < WAVE) break;
    }
#if REP_ACK_BYTES
    if (C.cs < C.pre_end) return;                                    /* (what was appended before this run commits first: rep_commit_pre) */
#endif
    /* ---- the ACKs: the (quorum - 1)-th largest of the followers' in-order counts ---- */
    uint64_t acked = ~0ull;
    if (quorum > 1) {
        const bool mem = lane < 16 && ((members >> lane) & 1u) && (pbw >> 40) == my_tag;
        uint64_t val = mem ? (pbw & PB_VAL) : 0ull;
#if REP_FAST_ACK
        if (lane < 16 && ((members >> lane) & 1u) && (pbf >> 40) == my_tag && (pbf & PB_VAL) > val) val = pbf & PB_VAL;
#endif
        uint32_t rank = 0;                                           /* values above this lane's (ties: the lower lane first) */
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint64_t vj = rl64u(val, j);
            rank += (vj > val || (vj == val && (uint32_t)j < lane)) ? 1u : 0u;
        }
        const unsigned long long sel = __ballot(lane < 16 && rank == quorum - 2);
        acked = sel ? rl64u(val, __builtin_ctzll(sel)) : 0ull;
    }
    uint64_t upto = acked < C.vis ? acked : C.vis;
    /* Entries this leader INHERITED (slots below the first entry of its own term, H_TERM_SLOT0 = the blank CONFIG entry of
     * become_leader) commit only BEHIND that entry: a follower's count says how much of the log it holds in order, not in
     * which term it came to hold it -- a voter that lags can hold the older entries without the CONFIG, and a majority by
     * count among such voters would commit older-term entries while this term's own entry sits on a minority: a later
     * leader may still overwrite them (the leader-completeness argument of Raft, figure 8; the reference never commits them
     * at all: its followers acknowledge to the dead sender, DESIGN 6 deviation 4).  So the commit does not move below
     * term_slot0 until the (quorum - 1)-th largest count covers the slot itself.   */
    if (C.cs < term_slot0 && acked <= term_slot0) upto = C.cs;
    if (upto > C.cs) { C.cs = upto; C.progress = true; }
}

/* what was appended before this run and is not committed yet (an exact-fit round left hidden, entries that had no
 * majority): the scan over the followers' per-entry ACK bytes */
__device__ static inline void rep_commit_pre(const EngDev &E, RepCommitState &C, const uint8_t *ackb, uint64_t cap, uint32_t members, uint32_t quorum)
{
    const uint64_t lim = C.vis < C.pre_end ? C.vis : C.pre_end;
    if (C.cs >= lim) return;
    RepAckWin<12, 1> win;
    rep_ack_load<12, 1>(win, ackb, cap, E.dir_mask, members, C.cs);
    const uint64_t upto = rep_ack_eval<12, 1>(win, E.dir_mask, members, quorum, C.cs, lim);
    if (upto > C.cs) { C.cs = upto; C.progress = true; }
}

__device__ static inline void rep_committer(const EngDev &E, const RepArgs &A, lds_u64 s_h, lds_u64 s_m, lds_u64 s_x)
{
    REP_SERIAL_PRIO();
    RepHost *H = A.H;
    RepLead *LS = A.LS;
    const RepDev &Md = E.rep[E.leader];
    const uint32_t lane = lane_id(), me = E.leader;
    const uint32_t size = E.group_size, size_mask = (1u << size) - 1, quorum = size / 2 + 1;
    const uint32_t members = size_mask & ~(1u << me);
    const uint32_t nf = (uint32_t)__popc(members);
    const uint8_t *ackb = E.ackb[me];
    const uint64_t cap = (uint64_t)E.dir_mask + 1;
    const RepBox *mybox = E.box[me];
    /* (lane f: the tag follower f's cumulative ACKs carry in this run -- its run counter as the host read it at the start) */
    const uint64_t my_tag = lane < APUS_DEV_MAX_SERVERS ? ((A.fruns[lane] + 1) & 0xFFFFFFull) : 0ull;
    RepCommitState C;
    C.t_done = 0;
    C.pre_end = s_h[H_N_END];
    C.vis = s_h[H_N_VISIBLE]; C.vis_off = s_h[H_END];
    C.n_end_seen = s_h[H_N_END];
    C.cs = s_h[H_N_COMMIT];
    C.slots_done = ld_sys(&H->slots_done);
    C.push_live = A.push_mask;
    const uint64_t term_slot0 = s_h[H_TERM_SLOT0];
    if (s_h[H_END] != E.log_len) C.vis = C.n_end_seen;   /* (an exact-fit round left hidden stays so until the next round is in) */
    else if (C.vis < C.n_end_seen) C.vis_off = Md.dir_off[(uint32_t)C.vis & E.dir_mask];
    uint64_t settled = ~0ull, cs_pub = C.cs, sd_pub = C.slots_done;
    uint64_t patience = 0;
    uint64_t st_pass = 0, st_prog = 0, st_busy = 0;
    const uint64_t st_t0 = wall_clock64();
    for (;;) {
        C.progress = false;
        st_pass++;
        const uint64_t st_p0 = (A.dbg & 512) ? wall_clock64() : 0;
        const uint64_t prog = s_m[M_PROG];
        const uint64_t tail = s_m[M_TAIL], fin = s_m[M_FINAL];
#if REP_ACK_BYTES
        if (C.cs < C.pre_end) rep_commit_pre(E, C, ackb, cap, members, quorum);
#endif
        rep_commit_pass(E, LS, C, tail, mybox, members, quorum, my_tag, term_slot0);
        if (lane == 0) { s_m[M_T_DONE] = C.t_done; s_m[M_CS] = C.cs; }      /* (the applier reads M_CS first) */
        if (C.slots_done != sd_pub) { sd_pub = C.slots_done; if (lane == 0) st_sys(&H->slots_done, C.slots_done + s_m[M_DROPPED]); }
        if (C.cs > cs_pub) {
            cs_pub = C.cs;
            if (lane == 0) { if (!(A.dbg & 2048)) st_sys(&H->commit_slot, C.cs); }
            else if (lane <= APUS_DEV_MAX_SERVERS && ((C.push_live >> (lane - 1)) & 1u)) st_sys(&E.box[lane - 1]->commit_bell, C.cs);   /* R4 */
        }
        /* ---- done? ---- */
        const uint64_t n_apply = s_m[M_N_APPLY];
        const bool can = (uint32_t)__popc((C.push_live | (1u << me)) & size_mask) >= quorum;
        /* (applied: everything committed -- or every TICKET, when entries from before the run committed in it: no ticket stands for
         *  those, the control-plane pass behind the run applies them) */
        const bool applied = n_apply == C.cs || (n_apply < C.pre_end && s_m[M_T_RETIRED] == C.t_done);
        if (C.t_done == tail && ((C.cs == C.vis && applied) || !can) && settled != prog) { settled = prog; if (lane == 0) st_sys(&H->settled, prog); }
        if (fin != ~0ull && C.t_done >= fin && C.cs == C.vis && applied) break;
        if (fin != ~0ull && C.t_done >= fin) {
            /* nothing more will be appended; ACKs may still be on their way -- unless no majority can answer */
            if (!can || ++patience > A.peer_polls) { if (can && lane == 0) spin_timeout(E, 7201); break; }
        } else if (fin != ~0ull && ++patience > 64 * A.peer_polls) { if (lane == 0) spin_timeout(E, 7202); break; }
        if (!C.progress) __builtin_amdgcn_s_sleep(1); else { st_prog++; if (A.dbg & 512) st_busy += wall_clock64() - st_p0; }
    }
    /* the applier takes what is committed, then the control words go back */
    if (lane == 0) { s_m[M_C_FINAL] = 1; LS->stat[1][0] = st_pass; LS->stat[1][1] = st_prog; LS->stat[1][2] = C.t_done; LS->stat[1][3] = wall_clock64() - st_t0; LS->stat[1][5] = st_busy; }
    for (uint64_t i = 0; !s_m[M_A_FINAL]; i++) {
        if (i > 64 * A.peer_polls) { if (lane == 0) spin_timeout(E, 7203); break; }
        __builtin_amdgcn_s_sleep(2);
    }
    const uint64_t n_apply = s_m[M_N_APPLY], ncl = s_m[M_A_NCL];
    const uint64_t c_off = (C.cs == C.vis) ? C.vis_off : ld_agent(&Md.dir_off[(uint32_t)C.cs & E.dir_mask]);
    const uint64_t a_off = (n_apply == C.vis) ? C.vis_off : ld_agent(&Md.dir_off[(uint32_t)n_apply & E.dir_mask]);
    if (lane == 0) {
        uint64_t *mh = Md.hdr;
        mh[H_N_VISIBLE] = C.vis;
        mh[H_COMMIT] = c_off; mh[H_N_COMMIT] = C.cs;
        mh[H_APPLY] = a_off; mh[H_N_APPLY] = n_apply;
        mh[H_APPLY_HASH] = s_h[H_APPLY_HASH] + s_m[M_A_HASH]; mh[H_APPLY_COUNT] = s_h[H_APPLY_COUNT] + ncl;
        mh[H_HIGHEST_REC] = s_h[H_HIGHEST_REC] + ncl;
        st_sys(&H->highest_rec, s_h[H_HIGHEST_REC] + ncl);
        st_sys(&H->commit_slot, C.cs);
        s_x[5] = C.cs;
    }
    /* what the followers acknowledged of the entries that did not commit (no majority): into the slot words
     * the control-plane kernels scan (k_control_round's commit_scan) */
    const uint64_t pbw = lane < 16 ? ld_sys(&mybox->persisted_by[lane]) : 0ull;
    const uint64_t held = (lane < 16 && (pbw >> 40) == my_tag) ? (pbw & PB_VAL) : 0ull;     /* (lane f: what follower f holds in order) */
    for (uint64_t s0 = C.cs; s0 < C.n_end_seen; s0 += WAVE) {      /* (the whole wavefront makes every pass: rl64u below) */
        const uint64_t s = s0 + lane;
        const bool in = s < C.n_end_seen;
        const uint32_t di = (uint32_t)s & E.dir_mask;
        const uint8_t want = rep_ack_tag(s, E.dir_mask);
        uint32_t bits = 0;
        for (uint32_t m = members; m; m &= m - 1) {
            const uint32_t f = (uint32_t)__builtin_ctz(m);
            const uint64_t hf = rl64u(held, (int)f);
#if REP_ACK_BYTES
            if (in && s < hf && ld_sys8(ackb + (uint64_t)f * cap + di) == want) bits |= 1u << f;
#else
            /* (an inherited entry counts as acknowledged by f only when f holds this term's first entry as well: above) */
            if (in && s < hf && (s >= term_slot0 || hf > term_slot0)) bits |= 1u << f;
#endif
        }
        if (!in) continue;
        __hip_atomic_store((APUS_GLOBAL uint32_t *)(uintptr_t)&Md.ack[di], bits, RLX_AGENT);
    }
    /* the last word on the commit, then the followers may park */
    if (lane >= 1 && lane <= APUS_DEV_MAX_SERVERS && ((C.push_live >> (lane - 1)) & 1u)) st_sys(&E.box[lane - 1]->commit_bell, C.cs);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

/* the leader applies round by round (it never runs do_action: highest_rec; apply_committed_entries,
 * dare_server.c:1815-1974): the apply records were written with the entries, here they are counted once their
 * round is committed */
__device__ static inline void rep_applier(const EngDev &E, const RepArgs &A, lds_u64 s_h, lds_u64 s_m)
{
    REP_SERIAL_PRIO();
    RepHost *H = A.H;
    RepLead *LS = A.LS;
    const uint32_t lane = lane_id();
    uint64_t t_app = 0, n_apply = s_h[H_N_APPLY], hash = 0, ncl = 0;
    const uint64_t hr0 = s_h[H_HIGHEST_REC];
    uint32_t lat_n = 0;
    uint64_t st_pass = 0, st_prog = 0, st_busy = 0;
    const uint64_t st_t0 = wall_clock64();
#if REP_APPLY_PRE
    bool pre_have = false, pre_lat = false, lone_ok = false;             /* the lone-round path's held granules: ticket pre_t's */
    uint32_t quiet = 0;
    uint64_t pre_t = 0, p1 = 0, p3 = 0, p4 = 0, p5 = 0, p6 = 0, p7 = 0;
#endif
    for (;;) {
        st_pass++;
        const uint64_t st_p0 = (A.dbg & 512) ? wall_clock64() : 0;
        /* (the committer's words as SCALARS: a value read from LDS is a vector value to the compiler, and every branch below that
         *  depends on one is compiled as a divergent one -- with the lone-round path in front of it the general path's loads ended
         *  up under exec masks, 3.2 -> 3.8 us per pass) */
        const uint64_t cfin = rl64u(s_m[M_C_FINAL], 0);
        const uint64_t cs = rl64u(s_m[M_CS], 0), t_done = rl64u(s_m[M_T_DONE], 0);
        bool progress = false;
#if REP_APPLY_PRE
        /* ---- rounds that come one by one (at most R_HOLD_MAX done tickets in front of the applier).  The commit is the last
         *      thing a lone round waits for, and its done granules were in memory long before (the append wavefront stored them
         *      when the bytes were in every ring; the committer counted the ticket as done then): this path asks for the next
         *      ticket's chunk of granules ONCE, holds it when that ticket's own are there, and from then on watches the
         *      committer's words in LDS alone -- when the commit comes the round is applied without another round trip (the
         *      general path below reads the commit first and loads second: a pass in flight fails on the commit it read before
         *      its loads and the next one loads again, 1.5 round trips behind the commit on average).  Granules are valid the
         *      moment they carry their ticket's tag and never change afterwards; lanes whose tickets were not done when the
         *      chunk was loaded fail their tag as they would have then, and are asked for again.  A path of its own, the general
         *      one untouched: with a queue of done tickets a held chunk is 64 rounds' worth of a pass that can take 512, and a
         *      branch inside the general path's loads cost it 0.8 us per pass (8.6 -> 7.9 G entries/s at one replica). ---- */
        /* (quiet: the applier found nothing to do R_QUIET passes in a row before this round came -- under load it never does, and there a
         *  catch-up of a few tickets taken one by one here, a round trip each, cost the general path 7 % at one replica) */
        if (t_app >= t_done) { if (++quiet >= R_QUIET) lone_ok = true; } else quiet = 0;
        if (lone_ok && t_app < t_done && t_done - t_app <= R_HOLD_MAX) {
            if (!(pre_have && pre_t == t_app)) {
                /* (the next ticket's granules alone, kept as scalars: six more vector registers alive across the general path's
                 *  96 cost it 0.6 us per pass) */
                const uint64_t ix = t_app % RS_CAP;
                uint64_t v1 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0, v7 = 0;
                pre_lat = lat_n < R_LAT_CAP && !(A.dbg & 2048);
                if (lane == 0) {
                    v1 = ld_agent(&LS->dn[DN_SLOT_END][ix]); v3 = ld_agent(&LS->dn[DN_HASH_LO][ix]); v4 = ld_agent(&LS->dn[DN_HASH_HI][ix]);
                    v5 = ld_agent(&LS->dn[DN_NCLIENT][ix]);
                    if (pre_lat) { v6 = ld_agent(&LS->dn[DN_T_APPENDED][ix]); v7 = ld_agent(&LS->dn[DN_T_SEQUENCED][ix]); }
                }
                p1 = rl64u(v1, 0); p3 = rl64u(v3, 0); p4 = rl64u(v4, 0); p5 = rl64u(v5, 0); p6 = rl64u(v6, 0); p7 = rl64u(v7, 0);
                pre_have = rep_gran_ok(p1, t_app) && rep_gran_ok(p3, t_app) && rep_gran_ok(p4, t_app) && rep_gran_ok(p5, t_app)
                           && (!pre_lat || (rep_gran_ok(p6, t_app) && rep_gran_ok(p7, t_app)));
                pre_t = t_app;
            }
            const uint64_t slot_end = rep_extend(n_apply, (uint32_t)p1);
            if (pre_have && slot_end <= cs) {
                const uint32_t nc = (uint32_t)p5;
                hash += (uint64_t)(uint32_t)p3 | (p4 << 32);
                ncl += nc;
                n_apply = slot_end;
                if (pre_lat) {
                    const uint32_t now = (uint32_t)wall_clock64(), t_seq = (uint32_t)p7, t_apd = (uint32_t)p6;
                    if (lane == 0 && t_seq && lat_n < R_LAT_CAP) { LS->lat_ticks[lat_n] = now - t_seq; LS->lat_app[lat_n] = now - t_apd; }
                    lat_n = min(lat_n + 1u, R_LAT_CAP);
                }
                t_app += 1;
                progress = true;
                pre_have = false;
                if (lane == 0) {
                    s_m[M_N_APPLY] = n_apply; s_m[M_T_RETIRED] = t_app;
                    if (nc && !(A.dbg & 2048)) st_sys(&H->highest_rec, hr0 + ncl);
                }
            }
        } else
#endif
        if (t_app < t_done) {
#if REP_APPLY_PRE
            lone_ok = false;
            t_app = rl64u(t_app, 0); n_apply = rl64u(n_apply, 0);      /* (scalars for the compiler: the lone-round path above updates them under a branch) */
#endif
            uint64_t g1[R_SUB], g3[R_SUB], g4[R_SUB], g5[R_SUB], g6[R_SUB], g7[R_SUB];
            const uint32_t nsub = (uint32_t)__builtin_amdgcn_readfirstlane((int)min((uint64_t)R_SUB, (t_done - t_app + WAVE - 1) / WAVE));       /* (as many chunks as rounds are done) */
            const bool want_lat = __builtin_amdgcn_readfirstlane((int)(lat_n < R_LAT_CAP && !(A.dbg & 2048))) != 0;         /* (the two time stamps only while latency samples are still being kept) */
#pragma unroll
            for (int s = 0; s < R_SUB; s++) {
                g1[s] = 0; g3[s] = 0; g4[s] = 0; g5[s] = 0; g6[s] = 0; g7[s] = 0;
                if ((uint32_t)s < nsub) {
                    const uint64_t ix = (t_app + (uint64_t)s * WAVE + lane) % RS_CAP;
                    g1[s] = ld_agent(&LS->dn[DN_SLOT_END][ix]); g3[s] = ld_agent(&LS->dn[DN_HASH_LO][ix]); g4[s] = ld_agent(&LS->dn[DN_HASH_HI][ix]);
                    g5[s] = ld_agent(&LS->dn[DN_NCLIENT][ix]);
                    if (want_lat) { g6[s] = ld_agent(&LS->dn[DN_T_APPENDED][ix]); g7[s] = ld_agent(&LS->dn[DN_T_SEQUENCED][ix]); }
                }
            }
            const uint64_t t0 = t_app;
            uint64_t nc_pass = 0, h_lane = 0;                          /* (summed over the lanes ONCE per pass, not once per chunk: three scans per chunk were a third of the pass) */
            uint32_t nc_lane = 0;
            const uint32_t now = (uint32_t)wall_clock64();             /* (once per pass: the latency samples' end) */
#pragma unroll
            for (int s = 0; s < R_SUB; s++) {
                if (t_app != t0 + (uint64_t)s * WAVE || (uint32_t)s >= nsub) break;
                const uint64_t k = t_app + lane;
                const uint64_t slot_end = rep_extend(n_apply, (uint32_t)g1[s]);
                const bool okk = k < t_done && rep_gran_ok(g1[s], k) && rep_gran_ok(g3[s], k) && rep_gran_ok(g4[s], k) && rep_gran_ok(g5[s], k)
                                 && (!want_lat || (rep_gran_ok(g6[s], k) && rep_gran_ok(g7[s], k))) && slot_end <= cs;
                const unsigned long long bal = __ballot(okk);
                const uint32_t p = (~bal) ? (uint32_t)__builtin_ctzll(~bal) : WAVE;
                if (!p) break;
                const bool mine = lane < p;
                if (mine) { h_lane += (uint64_t)(uint32_t)g3[s] | (g4[s] << 32); nc_lane += (uint32_t)g5[s]; }
                n_apply = rl64u(slot_end, (int)p - 1);
                const uint32_t t_seq = (uint32_t)g7[s], t_apd = (uint32_t)g6[s];
                if (want_lat && mine && t_seq && lat_n + lane < R_LAT_CAP) { LS->lat_ticks[lat_n + lane] = now - t_seq; LS->lat_app[lat_n + lane] = now - t_apd; }
                if (want_lat) lat_n = min(lat_n + p, R_LAT_CAP);
                t_app += p;
                progress = true;
            }
            if (progress) {
                hash += wsum64(h_lane); nc_pass = wsum32(nc_lane);
                ncl += nc_pass;
                if (lane == 0) {
                    s_m[M_N_APPLY] = n_apply; s_m[M_T_RETIRED] = t_app;
                    if (nc_pass && !(A.dbg & 2048)) st_sys(&H->highest_rec, hr0 + ncl);
                }
            }
        }
        if (cfin && !progress) break;                    /* (M_C_FINAL was read before M_CS / M_T_DONE: they were final) */
        if (!progress) __builtin_amdgcn_s_sleep(1); else { st_prog++; if (A.dbg & 512) st_busy += wall_clock64() - st_p0; }
    }
    if (lane == 0) {
        LS->lat_n = lat_n;
        LS->stat[2][5] = st_busy;
        LS->stat[2][0] = st_pass; LS->stat[2][1] = st_prog; LS->stat[2][2] = t_app; LS->stat[2][3] = wall_clock64() - st_t0;
        s_m[M_A_HASH] = hash; s_m[M_A_NCL] = ncl;
        s_m[M_A_FINAL] = 1;
    }
}

/* what the append wavefronts need of the engine descriptor, by group index, in LDS: an index that is not a
 * compile-time constant into the kernel's argument block is a global load (and a wait for every store in front of it) */
struct RepPtrLds {
    uint8_t *ring[APUS_DEV_MAX_SERVERS];
    RepBox  *box[APUS_DEV_MAX_SERVERS];
    uint64_t qbase[APUS_DEV_MAX_SERVERS];
};
struct RepAppLds {
    uint64_t pos[WAVE];
    uint64_t src[WAVE];
    uint32_t T[WAVE];
    uint32_t ubase[WAVE + 1];
    uint4    h0[WAVE];
    uint4    h1[WAVE];
#if REP_SPEC_PAY
    uint4    spec[WAVE];                 /* unit `lane`'s payload bytes as they came with the descriptors (kept here, not in registers, across the placement) */
#endif
};

/* reply bytes (dare_log_entry_t.reply[i] at byte 28 + i) of the servers in `mask4` (four of them) as the bytes of one word */
__device__ static inline uint32_t rep_spread4(uint32_t mask4) { return (mask4 & 1u) | ((mask4 & 2u) << 7) | ((mask4 & 4u) << 14) | ((mask4 & 8u) << 21); }
/* Reply bytes ride with the entry (round 4).  rc_send_entries_reply (dare_ibv_rc.c:1828-1863) leaves reply[f] = 1 in
 * follower f's own copy of an entry and in the sender's copy.  As separate one-byte stores from the follower's kernel those
 * were four read-modify-writes in HBM per entry at 3 replicas (a partial write-through store reads its sector first):
 * 128 B read + 128 B written of the 1016 B an entry moved.  A round that is one contiguous range of 16-byte units now
 * carries them in the header words the leader stores anyway: follower f's copy with reply[f], the leader's own with
 * reply[f] of every follower the round is pushed to -- the state every compared point shows.  What the bytes stand for
 * is unchanged: the commit is decided by the followers' cumulative in-order ACKs (persisted_by), never by these bytes.
 * The doorbell tells the follower that the bytes are there (R_BELL_REPLY); a follower that DECLINES an entry (a term older
 * than its own: the fence) clears its byte in both copies; a follower that is found behind the leader when the next run
 * starts (it died, or was dropped from the push set mid-run) has its byte cleared in the leader's copy of everything it
 * does not hold (k_rep_clear_reply, from apus_gpu_rep_start) before anything is caught up. */
#define R_BELL_REPLY (1u << 31)
#define R_BELL_META  (1u << 30)          /* granules 4..7 = idx of the round's first entry (lo, hi), its term (lo, hi); emeta[][] = the entries' clt_id / type / sender */

/* one append wavefront: ticket k, k + G, ... */
__device__ static inline void rep_append_wave(const EngDev &E, const RepArgs &A, RepAppLds &lds, const RepPtrLds &PT, uint32_t g, uint32_t G)
{
    RepHost *H = A.H;
    RepLead *LS = A.LS;
    const uint32_t lane = lane_id(), me = E.leader;
    const RepDev &Md = E.rep[me];
    const uint64_t L = E.log_len;
    const uint64_t term = Md.hdr[H_SID] >> 9;
    uint64_t a_rounds = 0, a_total = 0, a_drain = 0, a_desc = 0, a_pay = 0, a_pre = 0, a_it1 = 0, a_wait = 0, a_ldw = 0, a_iss = 0, a_sel = 0;
    uint64_t wv_next = 0;
    bool have_next = false;
    /* the passes without ticket words (PR_ALL64 ...: "wordless passes" above): the first one this wavefront has not left behind,
     * and its number modulo 65535 (the records' tags, kept up by addition: no 64-bit modulo per look) */
    uint64_t g_next = 0;
    uint32_t g_tag = 0;
    const uint32_t rj = lane >= 8 ? (lane - 8) >> 3 : 0u;               /* lanes 8..39: word (lane & 7) of record g_next + rj */
    /* one look: lanes 0..7 ticket kk's words, lanes 8..39 four records from g_next on -- ONE load instruction */
    auto look = [&](uint64_t kk) -> uint64_t {
        const uint64_t *p = lane < 8 ? &LS->tkw[lane][kk % RS_CAP] : &LS->grec[(g_next + rj) % GR_CAP][lane & 7];
        return lane < 40 ? ld_agent(p) : 0ull;
    };
    for (uint64_t k = g;; k += G) {
        /* ---- wait for ticket k: its eight words carry its tag, or a wordless pass holds it ---- */
        const uint64_t kx = k % RS_CAP;
        uint64_t wv = wv_next;                       /* (looked at under the previous round's store drain) */
        bool go = false, bulk = false, giant = false;
        int gfound = 0;
        const uint64_t t_top = (A.dbg & 256) ? wall_clock64() : 0;
        for (uint64_t i = 0;; i++) {
            /* after two looks that missed, only the word that is stored last and the first record are polled until one of
             * them carries its tag (a ticket's words sit in eight lines, the records in four more) */
            bool full = true;
            if (i >= 2 && (REP_POLL_WIDE == 0 || i < REP_POLL_WIDE)) {      /* (a wavefront that has waited long is waiting for a lone round: one look) */
                uint64_t m = 0;
                if (lane == 0) m = ld_agent(&LS->tkw[TK_META][kx]);
                else if (lane == 8) m = ld_agent(&LS->grec[g_next % GR_CAP][0]);
                full = rep_tk_ok(rl64u(m, 0), k) || (rl64u(m, 8) >> 48) == (uint64_t)g_tag + 1;
            }
            if (full && (i || !have_next)) wv = look(k);
            if (full) {
                const unsigned long long okb = __ballot(lane < 8 && rep_tk_ok(wv, k));
                /* a ticket of a bulk pass is its TK_META and TK_SRC words alone (looked at FIRST: the slot's other six words
                 * are whatever the last ticket that had eight left there) */
                const bool isb = ((okb >> TK_META) & 1ull) && (rdl64(wv, TK_META) & TK_BULK);
                if (isb && ((okb >> TK_SRC) & 1ull)) { go = true; bulk = true; break; }
                if (!isb && okb == 0xFFull) { go = true; break; }
                /* the wordless passes: the records whose every word carries the record's tag, in order; a pass whose tickets lie
                 * below k is left behind, the first one that does not decides */
                const uint32_t tg = g_tag + rj;
                const unsigned long long rb = __ballot(lane >= 8 && lane < 40 && (wv >> 48) == (uint64_t)((tg >= 65535u ? tg - 65535u : tg) + 1u));
                uint32_t adv = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (adv != (uint32_t)j || ((rb >> (8 + 8 * j)) & 0xFFull) != 0xFFull) break;
                    const uint64_t t0v = rdl64(wv, 8 + 8 * j + PR_T0) & TK_VAL, nv = (rdl64(wv, 8 + 8 * j + PR_BRF_N) >> 32) & 0x1FFFull;
                    if (k >= t0v + nv) { adv = (uint32_t)j + 1; continue; }
                    if (k >= t0v) { giant = true; gfound = j; }
                    break;
                }
                if (giant) { adv = (uint32_t)gfound; }
                g_next += adv; g_tag += adv; if (g_tag >= 65535u) g_tag -= 65535u;
                if (giant) { go = true; break; }
                if (adv == 4) { have_next = false; continue; }           /* (four passes left behind: the next four at once) */
            }
            if ((i & 7) == 7 && ld_agent(&LS->seq_final) <= k) break;      /* (one word for everybody: looked at now and then) */
            rep_nap(i < 256);
        }
        if (!go) {
            if (lane == 0 && a_rounds) {
                atomicAdd((unsigned long long *)&LS->stat[3][0], (unsigned long long)a_rounds);
                atomicAdd((unsigned long long *)&LS->stat[3][1], (unsigned long long)a_total);
                atomicAdd((unsigned long long *)&LS->stat[3][2], (unsigned long long)a_drain);
                atomicAdd((unsigned long long *)&LS->stat[3][3], (unsigned long long)a_desc);
                atomicAdd((unsigned long long *)&LS->stat[3][4], (unsigned long long)a_pay);
                atomicAdd((unsigned long long *)&LS->stat[3][5], (unsigned long long)a_pre);
                atomicAdd((unsigned long long *)&LS->stat[3][6], (unsigned long long)a_it1);
                atomicAdd((unsigned long long *)&LS->stat[3][7], (unsigned long long)a_wait);
                atomicAdd((unsigned long long *)&LS->stat[0][7], (unsigned long long)a_ldw);        /* (a free word of the sequencer's row) */
                atomicAdd((unsigned long long *)&LS->stat[4][4], (unsigned long long)a_iss);
                atomicAdd((unsigned long long *)&LS->stat[4][5], (unsigned long long)a_sel);
            }
            return;
        }
        const bool timed = A.dbg & 256;
        const bool nt_ring = A.dbg & 1024;            /* measurement: streaming ring stores + one release per round */
        const uint64_t t_seen = timed ? wall_clock64() : 0;
        a_wait += t_seen - t_top;
        /* a ticket that came with words: the slot does not keep them (their tag comes round again: "wordless passes" above) */
        if (!giant && lane == 0) st_agent(&LS->tkw[TK_META][kx], 0ull);
        wv &= TK_VAL;
        uint64_t e0, idx0, slot0, first, end_after, d0, d1, meta;
        uint32_t spec_T = 0;                          /* REP_SPEC_PAY: the one size of a round from the request ring, as its ticket / record says */
        ReqDev dbulk; dbulk.req_id = 0; dbulk.pay16_type = 0; dbulk.len = 0; dbulk.clt_id = 0;
        if (bulk || giant) {
            /* the ticket's eight words from the pass record and the staged prefix sums */
            uint64_t r = 0, bfirst = 0, pw = 0, pfv = 0;
            uint32_t bn = 0;
            bool bpin = false;
            if (giant) {
                /* the record is here (lanes 8 + 8 gfound ...): the round's number is the ticket's place in the pass */
                pw = rl64v(wv, (int)((lane & 7) + 8 + 8 * (uint32_t)gfound));            /* (lanes 0..7: the record's words, like a polled record) */
                const uint64_t t0 = rdl64(pw, PR_T0), rc0 = rdl64(pw, PR_RC0) & 0xFFFFFFFFull, brfn = rdl64(pw, PR_BRF_N);
                r = rc0 + (k - t0);
                if (brfn & PR_UNIFORM) { bn = (uint32_t)(rdl64(pw, PR_RC0) >> 32) & 0xFFu; bfirst = (brfn & 0xFFFFFFFFull) + (k - t0) * bn; }
                else {
                    /* rounds of different sizes: the round's first request is one more look (configs[3]) */
                    uint32_t fv = 0;
                    if (lane < 2) fv = E.round_first[r + lane];
                    bfirst = rl32u(fv, 0); bn = rl32u(fv, 1) - rl32u(fv, 0);
                }
                if (lane == 8) pfv = E.round_prefix[r]; else if (lane == 9) pfv = E.round_prefix[r + 1];
                if (lane < bn) dbulk = E.req[bfirst + lane];      /* (the round's descriptors: the same round trip) */
            } else {
                /* one round trip: the record's words and the round's four prefix values are asked for together */
                const uint64_t mv = rdl64(wv, TK_META);
                r = mv >> 16;
                const uint64_t p12 = mv & 0xFFFull;
                bpin = (mv & TK_BULK_PIN) != 0;                 /* a pass of full rounds from the request ring: r = the round's number within it */
                const uint64_t sv = rdl64(wv, TK_SRC);
                bfirst = sv & 0xFFFFFFFFull;
                bn = (uint32_t)(sv >> 32) & 0x7F;
                if (!bpin) {
                    if (lane == 8) pfv = E.round_prefix[r]; else if (lane == 9) pfv = E.round_prefix[r + 1];
                    if (lane < bn) dbulk = E.req[bfirst + lane];      /* (the round's descriptors: the same round trip) */
                }
                for (uint64_t i = 0;; i++) {
                    if (lane < 8) pw = ld_agent(&LS->prec[p12 % PR_CAP][lane]);
                    /* (the record was stored before the pass's tickets, but nothing orders their arrival) */
                    /* (valid: all eight words of ONE pass -- the same non-zero tag -- and that pass holds ticket k) */
                    {
                        const uint64_t tg = rdl64(pw, 0) >> 48, t0v = rdl64(pw, PR_T0) & TK_VAL, nv = ((rdl64(pw, PR_BRF_N) & TK_VAL) >> 32) & 0x1FFFull;
                        if (tg != 0 && __ballot(lane < 8 && (pw >> 48) == tg) == 0xFFull && k >= t0v && k - t0v < nv) break;
                    }
                    if (i > A.peer_polls) { if (lane == 0) spin_timeout(E, 7401); break; }
                    rep_nap(true);
                }
                pw &= TK_VAL;
            }
            const uint64_t t0 = rdl64(pw, PR_T0), rc0 = rdl64(pw, PR_RC0) & 0xFFFFFFFFull, end0 = rdl64(pw, PR_END0), pidx0 = rdl64(pw, PR_IDX0), pslot0 = rdl64(pw, PR_SLOT0);
            const uint64_t bpf = rdl64(pw, PR_BPF), brfn = rdl64(pw, PR_BRF_N), ps = rdl64(pw, PR_PUSH_STAMP);
            const uint64_t pf0 = rdl64(pfv, 8), pf1 = rdl64(pfv, 9), rf0 = bfirst, rf1 = bfirst + bn;
            const uint64_t brf = brfn & 0xFFFFFFFFull;
            if (bpin) {
                /* equally long entries, full rounds: everything is arithmetic on the round's number (bpf = the entries' size) */
                e0 = end0 + r * WAVE * bpf; idx0 = pidx0 + r * WAVE; slot0 = pslot0 + r * WAVE; first = bfirst;
                end_after = e0 + (uint64_t)WAVE * bpf; d0 = 0; d1 = ps & 0xFFFFFFFFull;
                meta = (uint64_t)WAVE | ((uint64_t)R_SRC_PINNED << 8) | ((ps >> 32) << 32);
                spec_T = (uint32_t)bpf;
                if (lane == 0 && (k - t0 != r || bfirst != ((brf + r * WAVE) & 0xFFFFFFFFull)) && !(atomicOr(E.status, 1u << 3) & (1u << 3))) { E.status[3] = (uint32_t)k; E.status[4] = (uint32_t)r; E.status[5] = (uint32_t)(k - t0); E.status[6] = 0xB01Du; }
            } else {
            e0 = end0 + (pf0 - bpf); idx0 = pidx0 + (rf0 - brf); slot0 = pslot0 + (rf0 - brf); first = rf0;
            end_after = end0 + (pf1 - bpf); d0 = 0; d1 = ps & 0xFFFFFFFFull;
            meta = (rf1 - rf0) | ((uint64_t)R_SRC_STAGED << 8) | ((ps >> 32) << 32);
            if (lane == 0 && rc0 + (k - t0) != r && !(atomicOr(E.status, 1u << 3) & (1u << 3))) { E.status[3] = (uint32_t)k; E.status[4] = (uint32_t)r; E.status[5] = (uint32_t)(rc0 + (k - t0)); E.status[6] = 0xB01Cu; }
            }
        } else {
            e0 = rdl64(wv, TK_E0); idx0 = rdl64(wv, TK_IDX0); slot0 = rdl64(wv, TK_SLOT0); first = rdl64(wv, TK_SRC);
            end_after = rdl64(wv, TK_END); d0 = rdl64(wv, TK_D0); d1 = rdl64(wv, TK_D1); meta = rdl64(wv, TK_META);
            if (((meta >> 8) & 0xF) == R_SRC_PINNED) spec_T = (uint32_t)d0;
        }
        const uint32_t n = (uint32_t)(meta & 0xFF), kind = (uint32_t)(meta >> 8) & 0xF, ctype = (uint32_t)(meta >> 16) & 0xFF;
        const uint32_t hidden = (uint32_t)(meta >> 12) & 1u;
        const uint32_t push = (uint32_t)(meta >> 32) & 0xFFFF;
        const uint32_t rings = ((A.dbg & 4) ? 0u : push) | (1u << me);
        const bool active = lane < n;
        /* ---- get_tailq_message: the requests of the round ---- */
        ReqDev d; d.req_id = 0; d.pay16_type = 0; d.len = 0; d.clt_id = 0;
        const uint8_t *src = nullptr;
        uint4 spec_v = make_uint4(0, 0, 0, 0);
        const bool spec_try = REP_SPEC_PAY && kind == R_SRC_PINNED && spec_T > APUS_HDR && spec_T - APUS_HDR <= R_INLINE && !(spec_T & 15u)
                              && ((spec_T >> 4) & ((spec_T >> 4) - 1)) == 0 && n * (spec_T >> 4) <= WAVE;      /* (a small round: <= 64 units of 16 bytes) */
        if (kind == R_SRC_PINNED) {
            /* descriptors and payload sit in host memory whose lines this CU may hold from an earlier, partly
             * filled state (a vector L1 is never refreshed by anybody's stores): drop them */
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
#if REP_SPEC_PAY
            /* the first 64 units' payload bytes, asked for with the descriptors: unit `lane` of a round whose entries are all spec_T
             * bytes long with their payload in the slot (what the loop below asks for in its first pass, when the descriptors agree) */
            if (spec_try) {
                const uint32_t s_upe = spec_T >> 4, s_so = 16u * (lane & (s_upe - 1)), s_e = lane >> __builtin_ctz(s_upe);
                if (s_so >= 48 && s_e < n) spec_v = ld16u((const uint8_t *)A.RQ->slot[(first + s_e) % RQ_CAP].pay + s_so - 50);
            }
#endif
            if (active) {
                const RepSlot *sl = &A.RQ->slot[(first + lane) % RQ_CAP];
                const uint4 q = *(const uint4 *)&sl->d;
                d.req_id = (uint64_t)q.x | ((uint64_t)q.y << 32); d.pay16_type = q.z; d.len = (uint16_t)(q.w & 0xFFFF); d.clt_id = (uint16_t)(q.w >> 16);
                src = (d.pay16_type & 0x0FFFFFFFu) == R_PAY_INLINE ? sl->pay : A.RQ->arena + (uint64_t)(d.pay16_type & 0x0FFFFFFFu) * 16;
            }
#if REP_SPEC_PAY
            if (spec_try) lds.spec[lane] = spec_v;          /* (loads return in order: it is here when the descriptors are) */
#endif
        } else if (kind == R_SRC_STAGED) {
            if (active) { if (bulk) d = dbulk; else d = E.req[first + lane]; src = E.arena + (uint64_t)(d.pay16_type & 0x0FFFFFFFu) * 16; }
        } else {
            d.pay16_type = ctype << 28;
        }
        const uint32_t T = active ? APUS_HDR + d.len : 0;
        const RepPlace pl = rep_place(e0, L, T, n);
        const uint64_t t_desc = timed ? wall_clock64() : 0;
        const uint64_t pos = rep_pos(pl, (int)lane);
        const uint64_t idx = rep_idx(pl, (int)lane, idx0);
        const uint32_t type = d.pay16_type >> 28;
        const uint32_t nu = active ? (T + 15) / 16 : 0;
        const uint32_t uincl = wscan32(nu);
        const uint4 h0 = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)term, (uint32_t)(term >> 32));
        const uint4 h1 = make_uint4((uint32_t)d.req_id, (uint32_t)(d.req_id >> 32), (uint32_t)d.clt_id | (type << 16) | (me << 24), 0);
        lds.pos[lane] = pos;
        lds.src[lane] = (uint64_t)(uintptr_t)src;
        lds.T[lane] = T;
        lds.ubase[lane] = uincl - nu;
        lds.h0[lane] = h0; lds.h1[lane] = h1;
        if (lane == WAVE - 1) lds.ubase[WAVE] = uincl;
        /* every entry of the same size?  (the followers then place the round from one number) */
        const uint32_t T0 = rl32u(T, 0);
        const bool uniform = !__ballot(active && T != T0);
        const bool bell_meta = kind != R_SRC_CONTROL && !(A.dbg & 16384);      /* client rounds: the followers need not read the headers back */
        const bool spec_ok = spec_try && uniform && T0 == spec_T && !__ballot(active && (d.pay16_type & 0x0FFFFFFFu) != R_PAY_INLINE);
        uint64_t mix = 0;
        uint32_t client = 0;
        uint4 ar0 = make_uint4(0, 0, 0, 0), ar1 = ar0;
        if (active) {
            const uint64_t slot = slot0 + lane;
            const uint32_t di = (uint32_t)slot & E.dir_mask;
            if (!(A.dbg & 8192)) {
            st_agent(&Md.dir_off[di], pos);
            __hip_atomic_store((APUS_GLOBAL uint32_t *)(uintptr_t)&Md.dir_len[di], T | (me << 24), RLX_AGENT);
            }
            /* the leader's apply record (apply_committed_entries, dare_server.c:1941-1955): written with the
             * entry, counted by the applier once the entry's round is committed */
            client = (type != APUS_NOOP && type != APUS_CONFIG && type != APUS_HEAD);
            /* (write-through like everything a run stores: one run laps the apply ring, and two XCDs' dirty copies of
             * one line could be written back in either order) */
            ar0 = make_uint4((uint32_t)slot, (uint32_t)(slot >> 32), (uint32_t)pos, (uint32_t)(pos >> 32));
            ar1 = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), T - APUS_HDR, (uint32_t)d.clt_id | (type << 16) | (client << 24));
            if (client) mix = apus_apply_mix(slot, pos, idx, T - APUS_HDR, d.clt_id, (uint8_t)type, 1);
            if (pl.stale && (int)lane == pl.kstar) {
                /* case 2 of the wrap: the header stays where it did fit (dare_log.h:521-538) */
                const uint4 z = make_uint4(0, 0, 0, 0), l4 = make_uint4((uint32_t)d.len, 0, 0, 0);
                for (uint32_t m = rings; m; m &= m - 1) {
                    uint8_t *rg = PT.ring[__builtin_ctz(m)];
                    st16_agent(rg + pl.a, h0); st16_agent(rg + pl.a + 16, h1); st16_agent(rg + pl.a + 32, z); st16_agent(rg + pl.a + 48, l4);
                }
            }
            if (!uniform) for (uint32_t m = push; m; m &= m - 1) {
                const uint32_t f = (uint32_t)__builtin_ctz(m);
                __hip_atomic_store((APUS_GLOBAL uint16_t *)(uintptr_t)&PT.box[f]->lens[(PT.qbase[f] + k) % RB_CAP][lane], (uint16_t)d.len, RLX_SYSTEM);
            }
            if (bell_meta && (!REP_BELL16 || n > 8)) for (uint32_t m = push; m; m &= m - 1) {       /* (<= 8 entries: the words ride in the doorbell) */
                const uint32_t f = (uint32_t)__builtin_ctz(m);
                __hip_atomic_store((APUS_GLOBAL uint32_t *)(uintptr_t)&PT.box[f]->emeta[(PT.qbase[f] + k) % RB_CAP][lane], h1.z, RLX_SYSTEM);
            }
        }
        if (!(A.dbg & 8192)) rep_store_rows32(ar0, ar1, n, [&](uint32_t r) { return (uint8_t *)&Md.apply[(uint32_t)(slot0 + r) & E.dir_mask]; });
        const uint64_t hsum = wsum64(mix);
        const uint32_t nclient = wsum32(client);
        {
            const uint64_t end_chk = rl64u(pos + T, (int)n - 1);        /* the sequencer and this wavefront must agree */
            const uint32_t T0x = rl32u(T, 0);
            if (lane == 0 && end_chk != end_after && !(atomicOr(E.status, 1u << 3) & (1u << 3))) {
                E.status[3] = (uint32_t)k; E.status[4] = (uint32_t)end_chk; E.status[5] = (uint32_t)end_after;
                E.status[6] = n | (kind << 8) | (T0x << 16); E.status[7] = (uint32_t)e0;
            }
        }
        /* ---- the round's bytes: own ring + R1 to every pushed follower ---- */
        const uint32_t utotal = lds.ubase[WAVE];
        const uint32_t upe = (T0 + 15) / 16;                       /* 16-byte units per entry when all are the same size */
        const uint64_t t_loop = timed ? wall_clock64() : 0;
        uint64_t t_it1 = 0;
        /* the common shape -- every entry the same multiple of 16 bytes, no wrap inside the round, 16-byte aligned --
         * is ONE contiguous range of bytes in every ring: unit u goes to e0 + 16 u */
        const bool straight = uniform && (T0 & 15u) == 0 && pl.kstar < 0 && (e0 & 15ull) == 0;
        const uint8_t *safe = (const uint8_t *)Md.dir_off;          /* where a lane that needs no payload bytes loads from */
        const bool preset = straight && !(A.dbg & (1 | 64));        /* the reply bytes ride with the entry */
        if (straight && (upe & (upe - 1)) == 0 && upe <= WAVE) {
            /* ... and with a power-of-two number of units per entry (64-, 192-, 448-byte payloads ...: configs[1]) a lane
             * writes the SAME 16 bytes of every entry it touches: which header word or payload bytes, the masks and the
             * length word are worked out once per round, not once per unit (a wavefront issues one instruction every four to
             * five cycles: the unit loop's divisions and masks were most of the round's 10 us) */
            constexpr int ILP = 8;
            const uint32_t P = T0 - APUS_HDR;
            const uint32_t sh = (uint32_t)__builtin_ctz(upe);
            const uint32_t so = 16u * (lane & (upe - 1));
            const bool is_ctl = kind == R_SRC_CONTROL;
            const bool is_pay = so >= 48 && !is_ctl && P != 0;
            uint64_t mlo = 0, mhi = 0, ins = 0;
            if (is_pay) {
                const int vb = so < 50 ? (int)(50 - so) : 0;
                const int ve = 50u + P > so ? (int)min(16u, 50u + P - so) : 0;
                mlo = byte_mask64(vb, ve); mhi = byte_mask64(vb - 8, ve - 8);
            }
            if (so == 48 && !is_ctl) ins = (uint64_t)(P & 0xFFFFu);
            const uint4 ctlv = make_uint4((uint32_t)d0, (uint32_t)(d0 >> 32), (uint32_t)d1, (uint32_t)(d1 >> 32));
#if REP_SPEC_PAY
            if (spec_ok) {
                /* a small round from the request ring (<= 64 units: spec_try) whose payload bytes came with its descriptors: unit
                 * `lane`, nothing to load, nothing to wait for */
                const uint32_t e = min(lane >> sh, n - 1);
                uint4 vv;
                if (so == 0) vv = lds.h0[e];
                else if (so == 16) vv = lds.h1[e];
                else if (so == 32) vv = make_uint4(0, 0, 0, 0);
                else {
                    const uint4 sv = lds.spec[lane];
                    const uint64_t lo = (((uint64_t)sv.x | ((uint64_t)sv.y << 32)) & mlo) | ins;
                    const uint64_t hi = ((uint64_t)sv.z | ((uint64_t)sv.w << 32)) & mhi;
                    vv = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
                }
                for (uint32_t m = rings; m; m &= m - 1) {
                    const uint32_t ri = (uint32_t)__builtin_ctz(m);
                    const uint32_t rmask = preset ? (ri == me ? push : (1u << ri)) : 0u;
                    uint4 rb = make_uint4(0, 0, 0, 0);
                    if (so == 16) rb.w = rep_spread4(rmask & 0xFu);
                    else if (so == 32) { rb.x = rep_spread4((rmask >> 4) & 0xFu); rb.y = rep_spread4((rmask >> 8) & 0xFu); rb.z = (rmask >> 12) & 1u; }
                    if (lane < utotal) st16_wt(PT.ring[ri] + e0 + 16ull * lane, make_uint4(vv.x | rb.x, vv.y | rb.y, vv.z | rb.z, vv.w | rb.w));
                }
            } else
#endif
            for (uint32_t u0 = lane; u0 < utotal; u0 += WAVE * ILP) {
                if (timed && u0 >= WAVE * ILP && !t_it1) t_it1 = wall_clock64();
                uint4 v[ILP], hv[ILP];
#pragma unroll
                for (int q = 0; q < ILP; q++) {
                    const uint32_t u = u0 + q * WAVE;
                    const uint32_t e = min(u >> sh, n - 1);
                    hv[q] = so == 0 ? lds.h0[e] : lds.h1[e];
                    v[q] = ld16u(is_pay && u < utotal ? (const uint8_t *)(uintptr_t)lds.src[e] + so - 50 : safe);
                }
                uint64_t tl1 = 0;
                if (timed) { const uint64_t tl0 = wall_clock64(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tl1 = wall_clock64(); a_ldw += tl1 - tl0; a_iss += tl0 - t_loop; }
#pragma unroll
                for (int q = 0; q < ILP; q++) {
                    if (so < 32) v[q] = hv[q];
                    else if (so == 32) v[q] = make_uint4(0, 0, 0, 0);
                    else if (is_ctl) v[q] = ctlv;
                    else {
                        const uint64_t lo = (((uint64_t)v[q].x | ((uint64_t)v[q].y << 32)) & mlo) | ins;
                        const uint64_t hi = ((uint64_t)v[q].z | ((uint64_t)v[q].w << 32)) & mhi;
                        v[q] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
                    }
                }
                if (timed) { asm volatile("" :: "v"(v[0].x), "v"(v[7].w)); a_sel += wall_clock64() - tl1; }
                for (uint32_t m = rings; m; m &= m - 1) {
                    const uint32_t ri = (uint32_t)__builtin_ctz(m);
                    uint8_t *rg = PT.ring[ri] + e0 + 16ull * u0;
                    const uint32_t rmask = preset ? (ri == me ? push : (1u << ri)) : 0u;
                    /* (this lane's unit of the entry: bytes 16..31 end with reply[0..3], bytes 32..47 begin with reply[4..12]) */
                    uint4 rb = make_uint4(0, 0, 0, 0);
                    if (so == 16) rb.w = rep_spread4(rmask & 0xFu);
                    else if (so == 32) { rb.x = rep_spread4((rmask >> 4) & 0xFu); rb.y = rep_spread4((rmask >> 8) & 0xFu); rb.z = (rmask >> 12) & 1u; }
#pragma unroll
                    for (int q = 0; q < ILP; q++) {
                        const uint32_t u = u0 + q * WAVE;
                        const uint4 vv = make_uint4(v[q].x | rb.x, v[q].y | rb.y, v[q].z | rb.z, v[q].w | rb.w);
                        if (u < utotal) { if (nt_ring) st16_nt(rg + 1024u * q, vv); else st16_wt(rg + 1024u * q, vv); }
                    }
                }
            }
        } else if (straight) {
            constexpr int ILP = 8;
            const uint32_t P = T0 - APUS_HDR;
            const uint32_t magic = 0xFFFFFFFFu / upe + 1u;           /* u / upe = (u * magic) >> 32 for u * upe < 2^32 */
            for (uint32_t u0 = lane; u0 < utotal; u0 += WAVE * ILP) {
                if (timed && u0 >= WAVE * ILP && !t_it1) t_it1 = wall_clock64();
                uint4 v[ILP];
                uint32_t so_[ILP], e_[ILP];
                /* the payload loads of ALL units of the pass go out together, for every lane, nothing but address arithmetic
                 * between them: ONE memory round trip per pass */
#pragma unroll
                for (int q = 0; q < ILP; q++) {
                    const uint32_t u = u0 + q * WAVE;
                    const uint32_t e = min(__umulhi(u, magic), n - 1);
                    e_[q] = e; so_[q] = 16u * (u - e * upe);
                    const bool need = u < utotal && so_[q] >= 48 && kind != R_SRC_CONTROL && P != 0;
                    v[q] = ld16u(need ? (const uint8_t *)(uintptr_t)lds.src[e] + so_[q] - 50 : safe);
                }
                if (timed) { const uint64_t tl0 = wall_clock64(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); a_ldw += wall_clock64() - tl0; }
#pragma unroll
                for (int q = 0; q < ILP; q++) {
                    if (so_[q] == 0) v[q] = lds.h0[e_[q]];
                    else if (so_[q] == 16) v[q] = lds.h1[e_[q]];
                    else if (so_[q] == 32) v[q] = make_uint4(0, 0, 0, 0);
                    else if (kind == R_SRC_CONTROL) v[q] = make_uint4((uint32_t)d0, (uint32_t)(d0 >> 32), (uint32_t)d1, (uint32_t)(d1 >> 32));
                    else v[q] = payload_mask(v[q], so_[q], P, P);
                }
                for (uint32_t m = rings; m; m &= m - 1) {
                    const uint32_t ri = (uint32_t)__builtin_ctz(m);
                    uint8_t *rg = PT.ring[ri] + e0;
                    const uint32_t rmask = preset ? (ri == me ? push : (1u << ri)) : 0u;
                    const uint32_t r16 = rep_spread4(rmask & 0xFu), r32x = rep_spread4((rmask >> 4) & 0xFu), r32y = rep_spread4((rmask >> 8) & 0xFu), r32z = (rmask >> 12) & 1u;
#pragma unroll
                    for (int q = 0; q < ILP; q++) {
                        const uint32_t u = u0 + q * WAVE;
                        uint4 vv = v[q];
                        if (so_[q] == 16) vv.w |= r16; else if (so_[q] == 32) { vv.x |= r32x; vv.y |= r32y; vv.z |= r32z; }
                        if (u < utotal) { if (nt_ring) st16_nt(rg + 16ull * u, vv); else st16_wt(rg + 16ull * u, vv); }
                    }
                }
            }
        } else {
            constexpr int ILP = 4;
            for (uint32_t u0 = lane; u0 < utotal; u0 += WAVE * ILP) {
                uint4 v[ILP];
                uint64_t p[ILP];
                uint32_t so_[ILP], e_[ILP];
                bool on[ILP];
#pragma unroll
                for (int q = 0; q < ILP; q++) {
                    const uint32_t u = u0 + q * WAVE;
                    on[q] = u < utotal;
                    uint32_t lo = 0, hi = n - 1;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi + 1) >> 1;
                        if (lds.ubase[mid] <= u) lo = mid; else hi = mid - 1;
                    }
                    const uint32_t e = lo, Te = lds.T[e], j = on[q] ? u - lds.ubase[e] : 0;
                    e_[q] = e;
                    so_[q] = min(16u * j, Te - 16u);
                    p[q] = lds.pos[e] + so_[q];
                    const bool need = on[q] && so_[q] >= 48 && kind != R_SRC_CONTROL && Te != APUS_HDR;
                    v[q] = ld16u(need ? (const uint8_t *)(uintptr_t)lds.src[e] + so_[q] - 50 : safe);
                }
#pragma unroll
                for (int q = 0; q < ILP; q++) {
                    if (!on[q]) continue;
                    const uint32_t Pq = lds.T[e_[q]] - APUS_HDR;
                    if (so_[q] == 0) v[q] = lds.h0[e_[q]];
                    else if (so_[q] == 16) v[q] = lds.h1[e_[q]];
                    else if (so_[q] == 32) v[q] = make_uint4(0, 0, 0, 0);
                    else if (kind == R_SRC_CONTROL) v[q] = make_uint4((uint32_t)d0, (uint32_t)(d0 >> 32), (uint32_t)d1, (uint32_t)(d1 >> 32));
                    else v[q] = payload_mask(v[q], so_[q], Pq, Pq);
                }
                for (uint32_t m = rings; m; m &= m - 1) {
                    uint8_t *rg = PT.ring[__builtin_ctz(m)];
#pragma unroll
                    for (int q = 0; q < ILP; q++)
                        if (on[q]) st16_agent(rg + p[q], v[q]);
                }
            }
        }
        const uint64_t t_stores = timed ? wall_clock64() : 0;
        /* the next ticket's words are asked for now: their round trip runs under the drain of this round's stores.
         * (Round 5 also asked for the next round's pass record, prefix sums and descriptors a round early, with this round's payload
         * loads -- 2.4 us of a round's 10.7 by the phase timers: the wait moved into the next phase, the throughput did not move
         * (3.62 / 3.49 against 3.73 / 3.55 G at three replicas): the sequencer hands out tickets at 14.8 ns per round, the append
         * wavefronts' capacity sits within 10 % of that -- shortening one of the two alone shows nothing.  Taken out
         * again.) */
        wv_next = look(k + G);
        have_next = true;
        if (nt_ring) rep_release(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t t_drained = timed ? wall_clock64() : 0;
        /* ---- the bytes are in every pushed ring.  R2: the round's doorbell in every pushed follower's mailbox;
         *      the round's done granules for the committer and the applier -- one batch of stores, no second drain ---- */
        const uint32_t t_now = (uint32_t)wall_clock64();
#if REP_BELL16
        const uint32_t em8 = (uint32_t)__shfl((int)h1.z, (int)((lane - 8) & 7u));     /* lanes 8..15: the emeta word of entry lane - 8 */
#else
        const uint32_t em8 = 0;
#endif
        if (lane < R_BELL_W) {
            const uint32_t val = lane == 0 ? (uint32_t)end_after : lane == 1 ? (uint32_t)(slot0 + n) : lane == 2 ? (uint32_t)e0
                                 : lane == 3 ? ((n << 17) | (uniform ? T0 : 0u) | (preset ? R_BELL_REPLY : 0u) | (bell_meta ? R_BELL_META : 0u))
                                 : !bell_meta ? 0u : lane == 4 ? (uint32_t)idx0 : lane == 5 ? (uint32_t)(idx0 >> 32) : lane == 6 ? (uint32_t)term
                                 : lane == 7 ? (uint32_t)(term >> 32) : em8;
            for (uint32_t m = push; m; m &= m - 1) {
                const uint32_t f = (uint32_t)__builtin_ctz(m);
                const uint64_t q = PT.qbase[f] + k;
                st_sys(&PT.box[f]->rnd[q % RB_CAP][lane], ((q + 1) << 32) | val);
            }
        }
        if (lane < 8) {
            uint32_t val = 0;
            switch (lane) {
            case DN_META: val = n | (kind << 8) | (hidden << 12) | (push << 16); break;
            case DN_SLOT_END: val = (uint32_t)(slot0 + n); break;
            case DN_END: val = (uint32_t)end_after; break;
            case DN_HASH_LO: val = (uint32_t)hsum; break;
            case DN_HASH_HI: val = (uint32_t)(hsum >> 32); break;
            case DN_NCLIENT: val = nclient; break;
            case DN_T_APPENDED: val = kind == R_SRC_CONTROL ? 0u : t_now; break;
            default: val = kind == R_SRC_CONTROL ? 0u : ((uint32_t)d1 ? (uint32_t)d1 : 1u); break;
            }
            st_agent(&LS->dn[lane][k % RS_CAP], rep_gran(k, val));
        }
        /* diagnostics: rounds and ticks from "ticket seen" to "done granules issued" */
        if (timed) { a_rounds++; a_total += wall_clock64() - t_seen; a_drain += t_drained - t_stores; a_desc += t_desc - t_seen; a_pay += t_stores - t_desc;
        a_pre += t_loop - t_desc; a_it1 += (t_it1 ? t_it1 : t_stores) - t_loop; }
    }
}

/* ===================================================================================== follower */
/* one work wavefront of follower `me`: rounds q0 + g, q0 + g + G, ... */
__device__ static inline void rep_follow_wave(const EngDev &E, const RepArgs &A, uint32_t me, uint32_t g, uint32_t G)
{
    RepBox *box = E.box[me];
    RepFollow *FS = A.FS[me];
    const RepDev &Md = E.rep[me];
    const uint32_t lane = lane_id();
    const uint64_t L = E.log_len;
    const uint64_t q0 = ld_sys(&box->f_seq_next), my_run = ld_sys(&box->f_runs);
    const uint64_t n_end0 = Md.hdr[H_N_END], my_sid = Md.hdr[H_SID];
    const uint32_t leader = E.leader;
    const uint64_t cap = (uint64_t)E.dir_mask + 1;
    uint8_t *const lring = leader < APUS_DEV_MAX_SERVERS ? E.rep[leader].ring : nullptr;     /* (once: the sender's log, ACK map and mailbox) */
    uint8_t *const lack = leader < APUS_DEV_MAX_SERVERS ? E.ackb[leader] : nullptr;
    RepBox *const lbox = leader < APUS_DEV_MAX_SERVERS ? E.box[leader] : nullptr;
    uint64_t bell_next = 0;
    bool have_bell = false;
#if REP_FAST_ACK
    uint64_t rp = 0;                          /* lanes 16, 17: the retire wavefront's two words as the doorbell's look saw them */
    const bool fast_on = !(A.dbg & 65536);    /* (APUS_REP_DBG & 65536: off at run time -- bench.py --gpus N sets it when first contact finds that a
                                               *  system-scope atomic into the peer's mailbox does not land before the store behind it, apus_selftest.h) */
    const uint64_t pb_tag = ((my_run + 1) & 0xFFFFFFull) << 40;
#endif
    const bool timed = A.dbg & 256;
    uint64_t w_rounds = 0, w_total = 0, w_hdr = 0, w_st = 0, w_drain = 0;
    for (uint64_t q = q0 + g;; q += G) {
        const uint32_t r = (uint32_t)(q % RB_CAP);
        /* ---- wait for the doorbell of round q ---- */
        uint64_t wv = bell_next;                     /* (looked at under the previous round's store drain) */
        uint32_t go = 0;
#if REP_FAST_ACK
        rp = 0;                                      /* (no granule carries tag 0) */
#endif
        for (uint64_t i = 0;; i++) {
            if (i || !have_bell) {
                if (lane < R_BELL_W) wv = ld_sys(&box->rnd[r][lane]);
#if REP_FAST_ACK
                /* (only by a wavefront that has been waiting: rounds that come one by one.  Under load the retire wavefront is
                 *  hundreds of rounds behind the doorbells and the second load per look cost 2-3 % at five and seven replicas) */
                if (fast_on && i >= 2 && lane >= 16 && lane < 18) rp = ld_agent(&FS->ret_pub[lane - 16]);      /* (issued BEHIND the doorbell's load: at least as new) */
#endif
            }
            if (__ballot(lane < R_BELL_W && (wv >> 32) == ((q + 1) & 0xFFFFFFFFull)) == (1ull << R_BELL_W) - 1) { go = 1; break; }      /* (the leader stores the whole line) */
            if ((i & 15) == 15) {
                const uint64_t ctrl = ld_sys(&box->ctrl);
                if ((ctrl >> 40) == my_run + 1 && q0 + (ctrl & 0xFFFFFFFFFFull) - 1 <= q) break;   /* parked: round q never comes */
                if (ld_agent(&FS->quit)) break;
            }
            rep_nap(i < 512);
        }
        if (!go) {
            if (timed && lane == 0 && w_rounds) {
                atomicAdd((unsigned long long *)&FS->stat[2][0], (unsigned long long)w_rounds);
                atomicAdd((unsigned long long *)&FS->stat[2][1], (unsigned long long)w_total);
                atomicAdd((unsigned long long *)&FS->stat[2][2], (unsigned long long)w_hdr);
                atomicAdd((unsigned long long *)&FS->stat[2][3], (unsigned long long)w_st);
                atomicAdd((unsigned long long *)&FS->stat[2][4], (unsigned long long)w_drain);
            }
            return;
        }
        const uint64_t t_bell = timed ? wall_clock64() : 0;
        const uint32_t end_after = (uint32_t)rdl64(wv, 0), slot_lo = (uint32_t)rdl64(wv, 1), e0 = (uint32_t)rdl64(wv, 2), w3 = (uint32_t)rdl64(wv, 3);
        const uint32_t n = (w3 >> 17) & 0x7Fu, Tu = w3 & 0x1FFFF;
        const bool preset = (w3 & R_BELL_REPLY) != 0;              /* the reply bytes came with the entries */
        const bool bmeta = (w3 & R_BELL_META) != 0 && !(A.dbg & 16);   /* ... and what the headers say came with the doorbell */
        const uint64_t b_idx0 = (uint64_t)(uint32_t)rdl64(wv, 4) | ((uint64_t)(uint32_t)rdl64(wv, 5) << 32);
        const uint64_t b_term = (uint64_t)(uint32_t)rdl64(wv, 6) | ((uint64_t)(uint32_t)rdl64(wv, 7) << 32);
#if REP_BELL16
        const uint32_t em_bell = (uint32_t)__shfl((int)(uint32_t)wv, (int)(8u + (lane & 7u)));      /* (every lane active here: the crossbar reads live lanes only) */
#endif
        /* the slot count's high half: this run stays within 2^31 slots of where it began */
        uint64_t slot_end = (n_end0 & ~0xFFFFFFFFull) | slot_lo;
        if (slot_end + (1ull << 31) < n_end0) slot_end += 1ull << 32;
        const uint64_t slot0 = slot_end - n;
        const bool active = lane < n;
        const uint32_t T = !active ? 0u : (Tu ? Tu : APUS_HDR + (uint32_t)__hip_atomic_load((const APUS_GLOBAL uint16_t *)(uintptr_t)&box->lens[r][lane], RLX_SYSTEM));
        const RepPlace pl = rep_place(e0, L, T, n);
        const uint64_t pos = rep_pos(pl, (int)lane);
        /* ---- persist_new_entries: the entries as they landed in the own log ---- */
        uint4 u0 = make_uint4(0, 0, 0, 0), u1 = u0;
        uint64_t mix = 0;
        uint32_t head_val = 0xFFFFFFFFu;
        uint32_t client = 0;
        uint4 ar0 = make_uint4(0, 0, 0, 0), ar1 = ar0;
        bool acked = false;
        uint64_t t_hdr = 0;
        if (active) {
            if (bmeta) {
                /* the header words this wavefront uses, from the doorbell and the round's 4 bytes per entry: idx (rep_idx over the
                 * same placement the leader made), term, clt_id / type / sender */
                const uint64_t ix = rep_idx(pl, (int)lane, b_idx0);
                u0 = make_uint4((uint32_t)ix, (uint32_t)(ix >> 32), (uint32_t)b_term, (uint32_t)(b_term >> 32));
#if REP_BELL16
                if (n <= 8) u1.z = em_bell;                                            /* (the doorbell's granules 8..15) */
                else
#endif
                u1.z = __hip_atomic_load((const APUS_GLOBAL uint32_t *)(uintptr_t)&box->emeta[r][lane], RLX_SYSTEM);
                if (A.dbg & 32768) {
                    /* verification mode (tests): the headers are read as well and must say the same */
                    uint4 c0, c1;
                    ld32_sys(Md.ring + pos, c0, c1);
                    if ((c0.x != u0.x || c0.y != u0.y || c0.z != u0.z || c0.w != u0.w || c1.z != u1.z) && !(atomicOr(E.status, 1u << 3) & (1u << 3))) {
                        E.status[3] = (uint32_t)q; E.status[4] = c0.x; E.status[5] = u0.x; E.status[6] = 0xBE11u; E.status[7] = c1.z ^ u1.z;
                    }
                }
            } else if (A.dbg & 16) ld32_dev(Md.ring + pos, u0, u1); else ld32_sys(Md.ring + pos, u0, u1);
            if (timed) t_hdr = wall_clock64();
            const uint64_t idx = (uint64_t)u0.x | ((uint64_t)u0.y << 32);
            const uint32_t type = (u1.z >> 16) & 0xFF, sender = u1.z >> 24;
            const uint16_t clt = (uint16_t)(u1.z & 0xFFFF);
            const uint64_t slot = slot0 + lane;
            const uint32_t di = (uint32_t)slot & E.dir_mask;
            /* rc_send_entries_reply (dare_ibv_rc.c:1828-1863): the own reply byte, R3 = the same byte in the
             * sender's log at the same offset, and the ACK byte in the sender's map -- unless this server has
             * moved on to a newer term than the one the round comes from (the term fence, receiver side).
             * (The ACK goes first: the entry IS in this log; what follows is this server's own bookkeeping.) */
            if (sender == leader && lring && (my_sid >> 9) <= (uint64_t)u0.z + ((uint64_t)u0.w << 32)) {
#if REP_ACK_BYTES
                st_sys8(lack + (uint64_t)me * cap + di, rep_ack_tag(slot, E.dir_mask));
#endif
                if (!preset && !(A.dbg & 1)) st_sys8(lring + pos + 28 + me, 1);
                acked = true;
            }
            if (!preset && !(A.dbg & 1)) st_sys8(Md.ring + pos + 28 + me, 1);
            if (preset && !acked && sender == leader && lring) st_sys8(lring + pos + 28 + me, 0);      /* declined: the sender's copy says so */
            if (!(A.dbg & 2)) {
            st_agent(&Md.dir_off[di], pos);
            __hip_atomic_store((APUS_GLOBAL uint32_t *)(uintptr_t)&Md.dir_len[di], T | (sender << 24), RLX_AGENT);
            }
            client = (type != APUS_NOOP && type != APUS_CONFIG && type != APUS_HEAD);
            ar0 = make_uint4((uint32_t)slot, (uint32_t)(slot >> 32), (uint32_t)pos, (uint32_t)(pos >> 32));
            ar1 = make_uint4(u0.x, u0.y, T - APUS_HDR, (uint32_t)clt | (type << 16) | ((client ? 2u : 0u) << 24));
            if (client) mix = apus_apply_mix(slot, pos, idx, T - APUS_HDR, clt, (uint8_t)type, 2);
            if (type == APUS_HEAD) { uint4 x0, x1; ld32_sys(Md.ring + pos + 32, x0, x1); head_val = x1.x; }
        }
        const uint32_t acked_all = __ballot(active && !acked) ? 0u : 1u;
#if REP_FAST_ACK
        {
            /* the retire wavefront stands in front of THIS round, and the round continues what it has retired: the cumulative
             * ACK it will send when the granules below reach it, sent now */
            const uint64_t rp0 = rdl64(rp, 16), rp1 = rdl64(rp, 17);
            if (lane == 0 && acked_all && lbox && rep_gran_ok(rp0, q) && rep_gran_ok(rp1, q) && (uint32_t)rp0 == e0 && (uint32_t)rp1 == (uint32_t)slot0
                && end_after != (uint32_t)L && n != 0)
                __hip_atomic_fetch_max((APUS_GLOBAL uint64_t *)(uintptr_t)&lbox->persisted_fast_by[me], pb_tag | (slot_end & PB_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
#endif
        if (!(A.dbg & 2)) rep_store_rows32(ar0, ar1, n, [&](uint32_t r) { return (uint8_t *)&Md.apply[(uint32_t)(slot0 + r) & E.dir_mask]; });
        const uint64_t hsum = wsum64(mix);
        const uint32_t nclient = wsum32(client);
        head_val = rl32u(head_val, 0);    /* (a <HEAD> entry is a round of its own) */
        /* ---- read, acknowledged entry by entry, the bookkeeping stores issued: the round's geometry for the retire wavefront,
         *      whose in-order count IS the ACK the leader commits on (R3) -- not held up by this server's own drain ---- */
        if (lane < 4) {
            const uint32_t val = lane == FR_END ? end_after : lane == FR_E0 ? e0 : lane == FR_SLOT_END ? (uint32_t)slot_end : (n | (nclient << 8) | (acked_all << 16));
            st_agent(&FS->fr[lane][r], rep_gran(q, val));
        }
        const uint64_t t_st = timed ? wall_clock64() : 0;
        if (lane < R_BELL_W) bell_next = ld_sys(&box->rnd[(q + G) % RB_CAP][lane]);     /* the next doorbell: its round trip runs under this drain */
        have_bell = true;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (timed) { const uint64_t t_dr = wall_clock64(); w_rounds++; w_total += t_dr - t_bell; w_hdr += rdl64(t_hdr, 0) - t_bell; w_st += t_st - rdl64(t_hdr, 0); w_drain += t_dr - t_st; }
        /* ---- everything this round stored is in memory: the granules the apply wavefront waits for ---- */
        if (lane >= FR_HASH_LO && lane <= FR_HEAD) {
            const uint32_t val = lane == FR_HASH_LO ? (uint32_t)hsum : lane == FR_HASH_HI ? (uint32_t)(hsum >> 32) : head_val;
            st_agent(&FS->fr[lane][r], rep_gran(q, val));
        }
    }
}

/* the follower's retire wavefront: rounds in order -- persist_new_entries' bookkeeping */
__device__ static inline void rep_follow_retire(const EngDev &E, const RepArgs &A, uint32_t me, lds_u64 s_f)
{
    REP_SERIAL_PRIO();
    RepBox *box = E.box[me];
    RepBox *lbox = E.box[E.leader];
    RepFollow *FS = A.FS[me];
    const RepDev &Md = E.rep[me];
    uint64_t *mh = Md.hdr;
    const uint32_t lane = lane_id();
    const uint64_t L = E.log_len;
    const uint64_t q0 = ld_sys(&box->f_seq_next), my_run = ld_sys(&box->f_runs);
    uint64_t end = mh[H_END], n_end = mh[H_N_END], store_count = mh[H_STORE_COUNT];
    const uint64_t my_sid = mh[H_SID];
    uint64_t q_ret = q0;
    /* an exact-fit round held back by the last run (its end is len: the log would read as empty) */
    uint64_t pend_n = 0, pend_slot_end = 0;
    {
        const uint64_t ps0 = ld_sys(&box->f_pend_slot0), pse = ld_sys(&box->f_pend_slot_end), psid = ld_sys(&box->f_pend_sid);
        if (pse > ps0 && psid == my_sid && ps0 == n_end) { pend_n = pse - ps0; pend_slot_end = pse; }
    }
    uint64_t real_n_end = n_end + pend_n;                /* slots consumed, the held-back round included */
    uint64_t real_end = pend_n ? L : end;
    const uint64_t pb_tag = ((my_run + 1) & 0xFFFFFFull) << 40;
    if (lane == 0) st_sys(&lbox->persisted_by[me], pb_tag | n_end);
#if REP_FAST_ACK
    if (lane == 0) st_sys(&lbox->persisted_fast_by[me], pb_tag | n_end);      /* (this run's tag; the work wavefronts only ever raise it) */
    if (lane < 2) st_agent(&FS->ret_pub[lane], pend_n ? 0ull : rep_gran(q_ret, lane == 0 ? (uint32_t)real_end : (uint32_t)real_n_end));
#endif
    uint64_t idle = 0;
    uint32_t exit_code = R_EXIT_STOP, last_p = 0;
    uint64_t final_q = ~0ull;
    uint64_t st_pass = 0, st_prog = 0, st_busy = 0;
    const uint64_t st_t0 = wall_clock64();
    RepFHost *FHm = A.FH[me];
    if (lane == 0 && FHm) { st_sys(&FHm->n_end, n_end); st_sys(&FHm->alive, 1); }
    for (;;) {
        bool progress = false;
        st_pass++;
        const uint64_t st_p0 = (A.dbg & 512) ? wall_clock64() : 0;
        const uint64_t ctrl = ld_sys(&box->ctrl);
        uint64_t f0[R_SUB], f1[R_SUB], f2[R_SUB], f3[R_SUB];
        /* (one chunk of 64 while the rounds trickle in, as many as the last look retired and two more once they come in bulk) */
        const uint32_t nsub = last_p < WAVE / 2 ? 1u : min((uint32_t)R_SUB, last_p / WAVE + 2);
#pragma unroll
        for (int s = 0; s < R_SUB; s++) {
            f0[s] = 0; f1[s] = 0; f2[s] = 0; f3[s] = 0;
            if ((uint32_t)s < nsub) {
                const uint64_t ix = (q_ret + (uint64_t)s * WAVE + lane) % RB_CAP;
                f0[s] = ld_agent(&FS->fr[FR_END][ix]); f1[s] = ld_agent(&FS->fr[FR_E0][ix]); f2[s] = ld_agent(&FS->fr[FR_SLOT_END][ix]); f3[s] = ld_agent(&FS->fr[FR_N][ix]);
            }
        }
        if (final_q == ~0ull && (ctrl >> 40) == my_run + 1) final_q = q0 + (ctrl & 0xFFFFFFFFFFull) - 1;
        const uint64_t t0 = q_ret;
        bool gap = false, fenced = false;
#pragma unroll
        for (int s = 0; s < R_SUB; s++) {
            if (q_ret != t0 + (uint64_t)s * WAVE || fenced || (uint32_t)s >= nsub) break;
            const uint64_t q = q_ret + lane;
            const bool okq = q < final_q && rep_gran_ok(f0[s], q) && rep_gran_ok(f1[s], q) && rep_gran_ok(f2[s], q) && rep_gran_ok(f3[s], q);
            const unsigned long long bal = __ballot(okq);
            uint32_t p = (~bal) ? (uint32_t)__builtin_ctzll(~bal) : WAVE;
            /* a round with an entry this server did not acknowledge (a term older than its own: the fence) ends the run
             * in front of it: the log stops where the deposed leader's rounds begin */
            const unsigned long long nak = __ballot(lane < p && !(((uint32_t)f3[s] >> 16) & 1u));
            if (nak) { p = min(p, (uint32_t)__builtin_ctzll(nak)); fenced = true; }
            if (!p) break;
            const bool mine = lane < p;
            const uint64_t ea = (uint32_t)f0[s], e0 = (uint32_t)f1[s], nn = (uint32_t)f3[s] & 0xFF;
            const uint64_t se = rep_extend(real_n_end, (uint32_t)f2[s]);
            /* the first round must continue where this log stands (a follower that missed rounds needs
             * the leader's catch-up first) */
            const uint64_t e0_0 = rl64u(e0, 0), s0_0 = rl64u(se - nn, 0);
            if (e0_0 != real_end || s0_0 != real_n_end) { gap = true; break; }
            /* ... and EVERY round continues the one before it: a doorbell that is not of this run's sequence (another
             * leader's, left behind in the ring) would carry this run's tag only by accident, never its geometry */
            {
                const uint32_t ea_p = wprev32((uint32_t)ea), se_p = wprev32((uint32_t)se);
                if (__ballot(mine && lane > 0 && ((uint32_t)e0 != ea_p || (uint32_t)(se - nn) != se_p))) { gap = true; break; }
            }
            /* rounds up to the last one that does not end on len become visible */
            const unsigned long long vis_b = __ballot(mine && ea != L);
            const uint64_t se_l = rl64u(se, (int)p - 1), ea_l = rl64u(ea, (int)p - 1);
            real_n_end = se_l; real_end = ea_l;
            if (vis_b) {
                const int jj = 63 - __builtin_clzll(vis_b);
                const uint64_t se_v = rl64u(se, jj), ea_v = rl64u(ea, jj);
                store_count += se_v - n_end;
                n_end = se_v; end = ea_v;
                pend_n = se_l - se_v; pend_slot_end = se_l;
            } else { pend_n = se_l - n_end; pend_slot_end = se_l; }
            q_ret += p;
            progress = true;
        }
        last_p = (uint32_t)(q_ret - t0);
        if (gap) { exit_code = R_EXIT_GAP; if (lane == 0) spin_timeout(E, 7301); break; }
#if REP_FAST_ACK
        if (progress && lane < 2) st_agent(&FS->ret_pub[lane], (pend_n || fenced) ? 0ull : rep_gran(q_ret, lane == 0 ? (uint32_t)real_end : (uint32_t)real_n_end));
#endif
        if (progress && lane == 0) {
            s_f[F_END] = end; s_f[F_N_END] = n_end;
            s_f[F_Q_RET] = q_ret;                       /* (the apply wavefront reads F_Q_RET first) */
            st_sys(&lbox->persisted_by[me], pb_tag | n_end);
            if (FHm && !(A.dbg & 2048)) st_sys(&FHm->n_end, n_end);
        }
        if (fenced) { exit_code = R_EXIT_FENCED; break; }
        /* the host asks this follower to leave (its leader is gone: nobody will ring the park doorbell) */
        if (!progress && FHm && final_q == ~0ull && (idle & 15) == 15 && ld_sys(&FHm->stop)) final_q = q_ret;
        /* ---- park? ---- */
        if (final_q != ~0ull && q_ret >= final_q) break;      /* everything that was sent is persisted */
        if (progress) { idle = 0; st_prog++; if (A.dbg & 512) st_busy += wall_clock64() - st_p0; }
        else {
            if (++idle > A.idle_polls) { exit_code = R_EXIT_IDLE; break; }
            rep_nap(idle < 64);
        }
    }
    if (lane == 0) {
        s_f[F_END] = end; s_f[F_N_END] = n_end; s_f[F_Q_RET] = q_ret;
        FS->stat[0][0] = st_pass; FS->stat[0][1] = st_prog; FS->stat[0][2] = q_ret - q0; FS->stat[0][3] = wall_clock64() - st_t0; FS->stat[0][5] = st_busy;
        s_f[F_STORE_COUNT] = store_count; s_f[F_PEND_N] = pend_n; s_f[F_PEND_SLOT_END] = pend_slot_end; s_f[F_EXIT] = exit_code;
        s_f[F_R_FINAL] = 1;
    }
}

/* the follower's apply wavefront: the commit doorbell (R4), apply_committed_entries round by round */
__device__ static inline void rep_follow_apply(const EngDev &E, const RepArgs &A, uint32_t me, lds_u64 s_f)
{
    REP_SERIAL_PRIO();
    RepBox *box = E.box[me];
    RepBox *lbox = E.box[E.leader];
    RepFollow *FS = A.FS[me];
    const RepDev &Md = E.rep[me];
    uint64_t *mh = Md.hdr;
    const uint32_t lane = lane_id();
    const uint64_t L = E.log_len;
    const uint64_t q0 = ld_sys(&box->f_seq_next), my_run = ld_sys(&box->f_runs);
    uint64_t head = mh[H_HEAD];
    uint64_t n_commit = mh[H_N_COMMIT], n_apply = mh[H_N_APPLY];
    const uint64_t c_off0 = mh[H_COMMIT], a_off0 = mh[H_APPLY];
    const uint64_t n_commit0 = n_commit, n_apply0 = n_apply;
    const uint64_t hash0 = mh[H_APPLY_HASH], cnt0 = mh[H_APPLY_COUNT], my_sid = mh[H_SID];
    uint64_t hash = 0, ncl = 0;
    uint64_t q_app = q0;
    /* (latched: a host consumer may register while the run is resident -- apus_gpu_rep_follower_replayed promises "before and
     * during a run" -- so the word is looked at again on every 256th pass (busy or idle: a look at pinned host memory is a PCIe round trip)
     * until it reads non-zero; round 4
     * read it once at the start, and a follower's FIRST term published the device's apply count alone: ADVICE r4) */
    bool has_consumer = A.FH[me] && ld_sys(&A.FH[me]->consumer) != 0;
    uint64_t applied_pub = n_apply;
    if (has_consumer) { const uint64_t hr = ld_sys(&A.FH[me]->replayed); if (hr < applied_pub) applied_pub = hr; }
    if (lane == 0) { st_sys(&lbox->seqdone_by[me], q_app); st_sys(&lbox->applied_by[me], applied_pub); st_sys(&lbox->apply_off_by[me], a_off0); }
    uint64_t idle_fin = 0;
    uint64_t end = 0, n_end = 0, q_ret = q0;
    uint64_t st_pass = 0, st_prog = 0, st_busy = 0;
    const uint64_t st_t0 = wall_clock64();
    for (;;) {
        bool progress = false;
        st_pass++;
        const uint64_t st_p0 = (A.dbg & 512) ? wall_clock64() : 0;
        const uint64_t rfin = s_f[F_R_FINAL];
        q_ret = s_f[F_Q_RET]; n_end = s_f[F_N_END]; end = s_f[F_END];
        uint64_t cs = ld_sys(&box->commit_bell);
        uint64_t f0[R_SUB], f2[R_SUB], f3[R_SUB], f4[R_SUB], f5[R_SUB], f6[R_SUB];
        const uint32_t nsub = (uint32_t)min((uint64_t)R_SUB, (q_ret - q_app + WAVE - 1) / WAVE);       /* (as many chunks as rounds are retired) */
        if (q_app < q_ret) {
#pragma unroll
            for (int s = 0; s < R_SUB; s++) {
                f0[s] = 0; f2[s] = 0; f3[s] = 0; f4[s] = 0; f5[s] = 0; f6[s] = 0;
                if ((uint32_t)s < nsub) {
                    const uint64_t ix = (q_app + (uint64_t)s * WAVE + lane) % RB_CAP;
                    f0[s] = ld_agent(&FS->fr[FR_END][ix]);
                    f2[s] = ld_agent(&FS->fr[FR_SLOT_END][ix]); f3[s] = ld_agent(&FS->fr[FR_N][ix]); f4[s] = ld_agent(&FS->fr[FR_HASH_LO][ix]);
                    f5[s] = ld_agent(&FS->fr[FR_HASH_HI][ix]); f6[s] = ld_agent(&FS->fr[FR_HEAD][ix]);
                }
            }
        }
        if (cs > n_end) cs = n_end;
        if (cs > n_commit) { n_commit = cs; progress = true; }
        if (q_app < q_ret) {
            const uint64_t t0 = q_app;
            uint64_t h_lane = 0;
            uint32_t nc_lane = 0;
#pragma unroll
            for (int s = 0; s < R_SUB; s++) {
                if (q_app != t0 + (uint64_t)s * WAVE || (uint32_t)s >= nsub) break;
                const uint64_t q = q_app + lane;
                const uint64_t se = rep_extend(n_apply, (uint32_t)f2[s]);
                const bool okq = q < q_ret && rep_gran_ok(f0[s], q) && rep_gran_ok(f2[s], q) && rep_gran_ok(f3[s], q) && rep_gran_ok(f4[s], q) && rep_gran_ok(f5[s], q)
                                 && rep_gran_ok(f6[s], q) && se <= n_commit && se <= n_end;
                const unsigned long long bal = __ballot(okq);
                const uint32_t p = (~bal) ? (uint32_t)__builtin_ctzll(~bal) : WAVE;
                if (!p) break;
                const bool mine = lane < p;
                if (mine) { h_lane += (uint64_t)(uint32_t)f4[s] | (f5[s] << 32); nc_lane += ((uint32_t)f3[s] >> 8) & 0xFF; }
                /* poll_config_entries: a committed <HEAD> entry moves the head (dare_server.c:2164) */
                const uint32_t hv = mine ? (uint32_t)f6[s] : 0xFFFFFFFFu;
                const unsigned long long hb = __ballot(hv != 0xFFFFFFFFu);
                /* ("larger" is relative to the log's end: the end of the <HEAD> entry's OWN round, as it was when the reference's
                 * follower looked at the entry -- not this server's persisted end of the moment, which may be most of a lap
                 * further on when the apply wavefront lags: the new head then reads as the older one and is never adopted) */
                /* (and EVERY <HEAD> entry of the chunk in its order: with many ticks in flight the heads of one pass can run
                 * around the ring -- the last one alone may read as older than a head of most of a lap ago) */
                for (unsigned long long b = hb; b; b &= b - 1) {
                    const int hl = __builtin_ctzll(b);
                    const uint64_t h = rl32u(hv, hl), ea_h = rl32u((uint32_t)f0[s], hl);
                    if (apus_is_larger(ea_h, L, h, head)) head = h;
                }
                n_apply = rl64u(se, (int)p - 1);
                q_app += p;
                progress = true;
            }
            if (q_app != t0) { hash += wsum64(h_lane); ncl += wsum32(nc_lane); }
            if (q_app != t0 && lane == 0) {
                if (!has_consumer) { st_sys(&lbox->applied_by[me], n_apply); applied_pub = n_apply; }
                st_sys(&lbox->seqdone_by[me], q_app);
                if (A.FH[me] && !(A.dbg & 2048)) st_sys(&A.FH[me]->n_apply, n_apply);
            }
        }
        if (!has_consumer && A.FH[me] && (st_pass & 255) == 0 && ld_sys(&A.FH[me]->consumer) != 0) {
            /* from here on the leader hears min(device apply, host replay); what it was told before stands (the head never moves back) */
            has_consumer = true;
        }
        /* a host consumer: applied = what the device has applied AND the host has replayed (a look at pinned host memory, only
         * while the two differ from what the leader was last told) */
        if (has_consumer && applied_pub < n_apply) {
            const uint64_t hr = ld_sys(&A.FH[me]->replayed);
            const uint64_t ap = hr < n_apply ? hr : n_apply;
            if (ap > applied_pub) { applied_pub = ap; if (lane == 0) st_sys(&lbox->applied_by[me], ap); }
        }
        /* ---- park?  everything that was sent is persisted; the leader's last commit doorbell was rung before the
         *      park word: one more look at it, then leave ---- */
        if (rfin) {
            if (!progress && (q_app == q_ret || ld_sys(&box->commit_bell) <= n_apply || idle_fin > 4)) break;
            if (!progress) idle_fin++;
        }
        if (!progress) rep_nap(true); else { st_prog++; if (A.dbg & 512) st_busy += wall_clock64() - st_p0; }
    }
    if (lane == 0) { FS->stat[1][0] = st_pass; FS->stat[1][1] = st_prog; FS->stat[1][2] = q_app - q0; FS->stat[1][3] = wall_clock64() - st_t0; FS->stat[1][5] = st_busy; }
    /* (F_R_FINAL was read before F_Q_RET / F_N_END / F_END in the last pass: they were final) */
    const uint32_t exit_code = (uint32_t)s_f[F_EXIT];
    const uint64_t c_off = n_commit == n_commit0 ? c_off0 : (n_commit == n_end ? end : ld_agent(&Md.dir_off[(uint32_t)n_commit & E.dir_mask]));
    const uint64_t a_off = n_apply == n_apply0 ? a_off0 : (n_apply == n_end ? end : ld_agent(&Md.dir_off[(uint32_t)n_apply & E.dir_mask]));
    if (lane == 0) {
        const uint64_t pend_n = s_f[F_PEND_N];
        mh[H_END] = end; mh[H_OLD_END] = end; mh[H_N_END] = n_end; mh[H_N_PERSIST] = n_end; mh[H_STORE_COUNT] = s_f[F_STORE_COUNT];
        mh[H_COMMIT] = c_off; mh[H_N_COMMIT] = n_commit;
        mh[H_APPLY] = a_off; mh[H_N_APPLY] = n_apply; mh[H_HEAD] = head;
        mh[H_APPLY_HASH] = hash0 + hash; mh[H_APPLY_COUNT] = cnt0 + ncl;
        st_agent(&FS->quit, 1ull);
        st_sys(&lbox->apply_off_by[me], a_off);
        /* the next run's rounds are numbered from a fresh stretch of the doorbell sequence: whatever a leader that
         * died (or was deposed) mid-flight left in rnd[] beyond q_ret -- it can have rung doorbells up to
         * q_app + RB_CAP - R_SLACK, out of order -- carries numbers no later run uses (tags must not be
         * reusable across runs and terms); the leader's rack[] granules are numbered with the same sequence */
        st_sys(&box->f_seq_next, q_ret + RB_CAP);
        st_sys(&box->f_pend_slot0, n_end); st_sys(&box->f_pend_slot_end, pend_n ? s_f[F_PEND_SLOT_END] : 0ull); st_sys(&box->f_pend_sid, my_sid);
        st_sys(&box->f_exit, (uint64_t)exit_code + 1);
        if (A.FH[me]) { st_sys(&A.FH[me]->n_apply, n_apply); st_sys(&A.FH[me]->exit_code, (uint64_t)exit_code); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_sys(&box->f_runs, my_run + 1);
        if (A.FH[me]) st_sys(&A.FH[me]->alive, 2);
        if (exit_code) atomicOr(E.status, 1u << 4);
    }
}

/* reply[f] = 0 in replica `lead`'s copy of the entry slots [s0, s1): follower f does not hold them (see R_BELL_REPLY).
 * Only where the slot is still a LIVE entry of the leader's log: the directory is sized for minimum-size entries and is
 * never invalidated by pruning, so a slot behind the head still resolves to a position -- one the ring may have lapped.
 * A slot counts as live when its position lies in [head, end) and the entry there carries the idx the slot must have
 * (idx runs back from the last entry's; a restart of the numbering at an exact fit in between leaves the bytes alone:
 * they are diagnostic state, the commit never looks at them).  ADVICE r4. */
__global__ __launch_bounds__(256) void k_rep_clear_reply(const EngDev E, uint32_t lead, uint32_t f, uint64_t s0, uint64_t s1,
                                                         uint64_t head, uint64_t end, uint64_t last_idx)
{
    const RepDev &Md = E.rep[lead];
    const uint64_t L = E.log_len;
    if (end == L) return;                                  /* the log reads as empty */
    const uint64_t d_head = apus_end_distance(end, L, head);
    for (uint64_t s = s0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < s1; s += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t pos = Md.dir_off[(uint32_t)s & E.dir_mask];
        if (pos + APUS_HDR > L) continue;
        const uint64_t d = apus_end_distance(end, L, pos);
        if (d == 0 || d > d_head) continue;                /* behind the head (or the end itself) */
        const uint64_t want = last_idx - (s1 - 1 - s);
        if (last_idx < s1 - 1 - s || ld8u(Md.ring + pos) != want) continue;
        st_sys8(Md.ring + pos + 28 + f, 0);
    }
}

/* ===================================================================================== the launch */
/* ONE RESIDENT LAUNCH PER PROCESS carries every role the process hosts.  A replica per GPU and process (BASELINE
 * configs[1] read literally: "a single persistent kernel per replica") therefore IS a kernel per replica, and that
 * case gets a kernel compiled for its role alone:
 *   k_replica_leader     the process hosts the leader and nothing else: control workgroup (sequencer, committer, applier) +
 *                        n_append append workgroups; the register budget of the sequencer / append wavefronts
 *   k_replica_follower   the process hosts exactly one follower: n_fwork workgroups (the first two wavefronts of the first
 *                        one are its retire and apply wavefronts), compiled for >= 4 wavefronts per SIMD (88 VGPRs)
 *   k_replica            several logical replicas in one process (the one-device bench and tests): [leader control,
 *                        n_append append workgroups,] then n_fwork workgroups per hosted follower, in ONE launch.
 * Round 5 measured the alternative for the last case -- a launch per logical replica, each on its own stream
 * (profiles/r05_split_launch_sweep.txt): 3.1-3.3 G entries/s against the single launch's 3.6-4.0 G at 3 replicas, and the
 * first run of a process did not finish (launches that wait for each other depend on the runtime mapping every stream
 * to a hardware queue of its own; it has four by default, 7 replicas would need seven).  Workgroups that wait for each
 * other belong in one launch.  All workgroups must be resident together: the host sizes the grid for that. */
#ifndef R_MIN_WG_PER_CU
#define R_MIN_WG_PER_CU 3      /* round 6: the sequencer's pass no longer holds a lane's worth of registers per round (233 -> 168 VGPRs): three
                                * workgroups per compute unit, 384 append workgroups instead of 192 (profiles/r06_grid_sweeps.txt) */
#endif
#ifndef R_FOLLOW_WAVES_PER_EU
#define R_FOLLOW_WAVES_PER_EU 4
#endif

/* workgroup b of the leader's 1 + n_append */
__device__ static inline void rep_leader_block(const EngDev &E, const RepArgs &A, const uint32_t b)
{
    __shared__ RepAppLds s_lds[4];
    __shared__ uint64_t s_h[64];
    __shared__ uint64_t s_ao[16];
    __shared__ uint64_t s_x[16];
    __shared__ uint64_t s_m[M_WORDS];
    __shared__ RepPtrLds s_pt;
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    if (b == 0) {
        RepHost *H = A.H;
        if (tid < 64) s_h[tid] = E.rep[E.leader].hdr[tid];
        if (tid < 16) s_x[tid] = 0;
        if (tid < M_WORDS) s_m[tid] = tid == M_FINAL ? ~0ull : (tid == M_N_APPLY ? E.rep[E.leader].hdr[H_N_APPLY] : 0ull);
        __syncthreads();
        if (wave == 0) rep_sequencer(E, A, APUS_LDS64(s_h), APUS_LDS64(s_ao), APUS_LDS64(s_m), APUS_LDS64(s_x), (uint4 *)&s_lds[0]);   /* (the append stage's LDS is free in this workgroup) */
        else if (wave == 1) rep_committer(E, A, APUS_LDS64(s_h), APUS_LDS64(s_m), APUS_LDS64(s_x));
        else if (wave == 2) rep_applier(E, A, APUS_LDS64(s_h), APUS_LDS64(s_m));
        __syncthreads();
        if (tid == 0) {
            /* park the followers: every round they were sent is in their doorbell ring */
            const uint64_t t_fin = s_x[1];
            for (uint32_t m = A.park_mask; m; m &= m - 1) {
                const uint32_t f = (uint32_t)__builtin_ctz(m);
                uint64_t sent = 0;
                if ((A.push_mask >> f) & 1u) { const uint64_t td = ld_agent(&A.LS->t_drop[f]); sent = td == ~0ull ? t_fin : td; }
                st_sys(&E.box[f]->ctrl, ((A.fruns[f] + 1) << 40) | (sent + 1));
            }
            st_sys(&H->exit_code, s_x[0]);
            st_sys(&H->rounds, t_fin);
            if (s_x[0] == R_EXIT_TIMEOUT) atomicOr(E.status, 1u << 4);
            __threadfence_system();
            st_sys(&H->alive, 2);
        }
        return;
    }
    if (tid < APUS_DEV_MAX_SERVERS) { s_pt.ring[tid] = E.rep[tid].ring; s_pt.box[tid] = E.box[tid]; s_pt.qbase[tid] = A.qbase[tid]; }
    __syncthreads();
    rep_append_wave(E, A, s_lds[wave], s_pt, (b - 1) * 4 + wave, A.n_append * 4);
}

/* workgroup fb of follower me's n_fwork */
__device__ static inline void rep_follower_block(const EngDev &E, const RepArgs &A, const uint32_t me, const uint32_t fb)
{
    __shared__ uint64_t s_f[F_WORDS];
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    const uint32_t G = A.n_fwork * 4 - 2;
    if (fb == 0) {
        if (tid < F_WORDS) {
            const uint64_t *mh = E.rep[me].hdr;
            s_f[tid] = tid == F_END ? mh[H_END] : tid == F_N_END ? mh[H_N_END] : tid == F_Q_RET ? ld_sys(&E.box[me]->f_seq_next) : 0ull;
        }
        __syncthreads();
        if (wave == 0) { rep_follow_retire(E, A, me, APUS_LDS64(s_f)); return; }
        if (wave == 1) { rep_follow_apply(E, A, me, APUS_LDS64(s_f)); return; }
        rep_follow_wave(E, A, me, wave - 2, G);
        return;
    }
    rep_follow_wave(E, A, me, fb * 4 + wave - 2, G);
}

/* the XCD this workgroup runs on (HW_REG_XCC_ID = hwreg 20, bits [3:0]) */
__device__ static inline uint32_t rep_xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }
__device__ static inline void rep_note_xcc(const RepArgs &A)
{
    if (A.LS && threadIdx.x == 0 && blockIdx.x < 1024) A.LS->xcc[blockIdx.x] = (uint8_t)(rep_xcc_id() + 1);
}

__global__ __launch_bounds__(256, R_MIN_WG_PER_CU) void k_replica(const EngDev E, const RepArgs A)
{
    rep_note_xcc(A);
    uint32_t b = blockIdx.x;
    if (A.lead_here) {
        if (b <= A.n_append) { rep_leader_block(E, A, b); return; }
        b -= 1 + A.n_append;
    }
    const uint32_t ord = b / A.n_fwork, fb = b % A.n_fwork;
    int me = -1;
    for (int i = 0, k = 0; i < APUS_DEV_MAX_SERVERS; i++)
        if (A.follow_mask & (1u << i)) { if (k == (int)ord) { me = i; break; } k++; }
    if (me < 0) return;
    rep_follower_block(E, A, (uint32_t)me, fb);
}

__global__ __launch_bounds__(256, R_MIN_WG_PER_CU) void k_replica_leader(const EngDev E, const RepArgs A)
{
    rep_note_xcc(A);
    rep_leader_block(E, A, blockIdx.x);
}

__global__ __launch_bounds__(256, R_FOLLOW_WAVES_PER_EU) void k_replica_follower(const EngDev E, const RepArgs A, const uint32_t me)
{
    rep_follower_block(E, A, me, blockIdx.x);
}


/* ===================================================================================== link calibration */
/* The reference measures its fabric before it runs (rc_get_loggp_params, dare_ibv_rc.c:3323-3739: overhead, latency
 * and gap per byte of RDMA WRITE / READ); these two kernels do the same for the path the replica kernels use --
 * system-scope stores into a peer's HBM (xGMI between two GPUs, the device's own fabric between two processes on one):
 *   k_calib_pingpong   one 8-byte store to the peer's mailbox, the peer's kernel answers with one: the round trip of a
 *                      doorbell, `iters` times, every sample kept.  role 0 starts, role 1 answers.
 *   k_calib_store      every workgroup streams write-through 16-byte stores into the peer's ring (the R1 push). */
__global__ __launch_bounds__(64) void k_calib_pingpong(const EngDev E, uint32_t me, uint32_t peer, uint32_t role, uint32_t iters,
                                                     uint64_t base, uint32_t *ticks, uint64_t max_polls)
{
    RepBox *mine = E.box[me], *theirs = E.box[peer];
    if (threadIdx.x != 0) return;
    for (uint32_t i = 1; i <= iters; i++) {
        const uint64_t want = base + i;
        if (role == 0) {
            const uint64_t t0 = wall_clock64();
            st_sys(&theirs->ping, want);
            uint64_t polls = 0;
            while (ld_sys(&mine->pong) != want) if (++polls > max_polls) { ticks[0] = 0xFFFFFFFFu; return; }
            ticks[i] = (uint32_t)(wall_clock64() - t0);
        } else {
            uint64_t polls = 0;
            while (ld_sys(&mine->ping) != want) if (++polls > max_polls) { ticks[0] = 0xFFFFFFFFu; return; }
            st_sys(&theirs->pong, want);
        }
    }
    ticks[0] = iters;
}

__global__ __launch_bounds__(256) void k_calib_store(uint8_t *dst, uint64_t bytes, uint32_t seed)
{
    const uint64_t units = bytes / 16;
    const uint4 v = make_uint4(seed, blockIdx.x, threadIdx.x, 0x9E3779B9u);
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (uint64_t)gridDim.x * blockDim.x)
        st16_wt(dst + 16 * u, v);
}
