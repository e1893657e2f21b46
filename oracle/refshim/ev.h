/*
 * ev.h -- deterministic stand-in for libev.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference's dare_server.c (under /root/reference, compiled unmodified) arms four
 * ev_timers and one ev_idle watcher and then blocks in ev_run().  For a replayable
 * oracle the event loop must not depend on wall-clock time, so this header provides the
 * handful of libev names the reference uses with harness-driven semantics:
 *   - ev_run() returns at once (the harness owns the loop);
 *   - a timer only records {armed, repeat, callback}; oracle/refshim/glue.c exposes
 *     "fire timer X now" and "run the idle watcher once" to the trace driver;
 *   - ev_now() is a virtual clock the harness advances.
 * API names follow libev's public documentation; no libev code is used.
 */
#ifndef APUS_FAKE_EV_H
#define APUS_FAKE_EV_H

typedef double ev_tstamp;

struct ev_loop { ev_tstamp now; int broken; };

#define EV_P        struct ev_loop *loop
#define EV_P_       EV_P,
#define EV_A        loop
#define EV_A_       EV_A,
#define EV_MAXPRI   2
#define EV_MINPRI   (-2)
#define EVBREAK_ALL 2

typedef struct ev_timer {
    int active;
    int priority;
    ev_tstamp at;
    ev_tstamp repeat;
    void (*cb)(struct ev_loop *loop, struct ev_timer *w, int revents);
    void *data;
} ev_timer;

typedef struct ev_idle {
    int active;
    int priority;
    void (*cb)(struct ev_loop *loop, struct ev_idle *w, int revents);
    void *data;
} ev_idle;

struct ev_loop *apus_ev_default_loop(void);
#define EV_DEFAULT apus_ev_default_loop()

#define ev_timer_init(w, cb_, after_, repeat_) \
    do { (w)->active = 0; (w)->priority = 0; (w)->cb = (cb_); (w)->at = (after_); (w)->repeat = (repeat_); } while (0)
#define ev_idle_init(w, cb_) do { (w)->active = 0; (w)->priority = 0; (w)->cb = (cb_); } while (0)
#define ev_set_cb(w, cb_)    ((w)->cb = (cb_))
#define ev_set_priority(w, p) ((w)->priority = (p))

void ev_timer_again(struct ev_loop *loop, ev_timer *w);
void ev_timer_stop(struct ev_loop *loop, ev_timer *w);
void ev_idle_start(struct ev_loop *loop, ev_idle *w);
ev_tstamp ev_now(struct ev_loop *loop);
int  ev_run(struct ev_loop *loop, int flags);
void ev_break(struct ev_loop *loop, int how);

#endif
