/* db_stub.c -- the flat-file record store behind refshim/proc/db.h.  TEST INFRASTRUCTURE ONLY. */
#include <errno.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include "db.h"

static int s_pagesize(DB *d, uint32_t n) { d->pagesize = n ? n : 4096; return 0; }
static int s_cachesize(DB *d, uint32_t g, uint32_t b, int n) { (void)d; (void)g; (void)b; (void)n; return 0; }
static int s_open(DB *d, DB_TXN *t, const char *file, const char *name, DBTYPE type, uint32_t flags, int mode)
{
    (void)t; (void)name; (void)type; (void)flags; (void)mode;
    d->f = file ? fopen(file, "w+b") : NULL;
    if (file && !d->f) return errno ? errno : EIO;
    if (d->f) setvbuf(d->f, NULL, _IONBF, 0);           /* the page buffer below is the cache */
    d->page = malloc(d->pagesize);
    return d->page ? 0 : ENOMEM;
}
static void flush_page(DB *d) { if (d->f && d->page_used) fwrite(d->page, 1, d->page_used, d->f); d->page_used = 0; }
static int s_put(DB *d, DB_TXN *t, DBT *key, DBT *data, uint32_t flags)
{
    (void)t; (void)flags;
    const uint32_t sz = data->size;
    /* the page cache: records go into the current page, a full page goes to the file */
    uint32_t off = 0;
    while (off < sz) {
        uint32_t room = d->pagesize - d->page_used, k = sz - off < room ? sz - off : room;
        memcpy(d->page + d->page_used, (const uint8_t *)data->data + off, k);
        d->page_used += k; off += k;
        if (d->page_used == d->pagesize) flush_page(d);
    }
    /* ... and the records themselves, for the cursor */
    if (d->recs_len + 4 + sz > d->recs_cap) {
        uint64_t nc = d->recs_cap ? d->recs_cap * 2 : (1u << 20);
        while (nc < d->recs_len + 4 + sz) nc *= 2;
        uint8_t *p = realloc(d->recs, nc);
        if (!p) return ENOMEM;
        d->recs = p; d->recs_cap = nc;
    }
    memcpy(d->recs + d->recs_len, &sz, 4);
    memcpy(d->recs + d->recs_len + 4, data->data, sz);
    d->recs_len += 4 + sz;
    d->last_key = (db_recno_t)++d->n;
    if (key) {
        if (key->flags & DB_DBT_MALLOC) { key->data = malloc(sizeof(db_recno_t)); if (key->data) memcpy(key->data, &d->last_key, sizeof(db_recno_t)); }
        key->size = sizeof(db_recno_t);
    }
    return 0;
}
static int s_get(DB *d, DB_TXN *t, DBT *k, DBT *v, uint32_t f) { (void)d; (void)t; (void)k; (void)v; (void)f; return DB_NOTFOUND; }
static int s_sync(DB *d, uint32_t f) { (void)f; flush_page(d); if (d->f) fflush(d->f); return 0; }
static int s_close(DB *d, uint32_t f) { (void)f; flush_page(d); if (d->f) fclose(d->f); free(d->page); free(d->recs); free(d); return 0; }
static void s_err(DB *d, int e, const char *fmt, ...)
{
    (void)d;
    va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
    fprintf(stderr, ": %s\n", db_strerror(e));
}
static int c_get(DBC *c, DBT *key, DBT *data, uint32_t flags)
{
    (void)flags;
    DB *d = c->db;
    if (c->pos >= d->recs_len) return DB_NOTFOUND;
    uint32_t sz; memcpy(&sz, d->recs + c->pos, 4);
    data->data = d->recs + c->pos + 4; data->size = sz;
    c->pos += 4 + sz; c->key++;
    if (key) { key->data = &c->key; key->size = sizeof c->key; }
    return 0;
}
static int c_close(DBC *c) { free(c); return 0; }
static int s_cursor(DB *d, DB_TXN *t, DBC **out, uint32_t f)
{
    (void)t; (void)f;
    DBC *c = calloc(1, sizeof *c);
    if (!c) return ENOMEM;
    c->c_get = c->get = c_get; c->c_close = c->close = c_close; c->db = d;
    *out = c;
    return 0;
}
int db_create(DB **out, DB_ENV *env, uint32_t flags)
{
    (void)env; (void)flags;
    DB *d = calloc(1, sizeof *d);
    if (!d) return ENOMEM;
    d->set_pagesize = s_pagesize; d->set_cachesize = s_cachesize; d->open = s_open; d->close = s_close; d->put = s_put;
    d->get = s_get; d->cursor = s_cursor; d->sync = s_sync; d->err = s_err; d->pagesize = 4096;
    *out = d;
    return 0;
}
char *db_strerror(int e) { return e == DB_NOTFOUND ? "DB_NOTFOUND: no matching key/data pair" : strerror(e); }
