/*
 * db.h -- flat-file stand-in for the BerkeleyDB names src/db/db-interface.c uses.  TEST INFRASTRUCTURE ONLY
 * (BASELINE configs[0], the reference-as-is baseline: oracle/Makefile target `procref`).
 *
 * The reference stores every persisted entry as one record of a DB_RECNO database (db-interface.c:22-105) and walks
 * them with a cursor for a snapshot (:107-135).  BerkeleyDB's headers are not in this image; its on-disk format is
 * off the path.  What is kept: one put() = one record appended (memcpy into a 32 KiB page buffer -- the reference's
 * pagesize -- flushed to the file when it fills, like a write-back page cache), cursor walk in insertion order.
 * Names and call shapes follow the BerkeleyDB C API documentation; no BerkeleyDB code is used.
 */
#ifndef APUS_FAKE_DB_H
#define APUS_FAKE_DB_H
#include <stdint.h>
#include <stdio.h>
#include <sys/types.h>

typedef struct __db DB;
typedef struct __dbc DBC;
typedef struct __db_txn DB_TXN;
typedef struct __db_env DB_ENV;
typedef uint32_t db_recno_t;

typedef struct __dbt {
    void *data;
    uint32_t size, ulen, dlen, doff;
    void *app_data;
    uint32_t flags;
} DBT;

typedef enum { DB_BTREE = 1, DB_HASH = 2, DB_RECNO = 3, DB_QUEUE = 4, DB_UNKNOWN = 5 } DBTYPE;

#define DB_CREATE       0x00000001
#define DB_THREAD       0x00000020
#define DB_AUTO_COMMIT  0x00000100
#define DB_APPEND       2
#define DB_NEXT         16
#define DB_DBT_MALLOC   0x010
#define DB_NOTFOUND     (-30988)

struct __db {
    int (*set_pagesize)(DB *, uint32_t);
    int (*set_cachesize)(DB *, uint32_t, uint32_t, int);
    int (*open)(DB *, DB_TXN *, const char *, const char *, DBTYPE, uint32_t, int);
    int (*close)(DB *, uint32_t);
    int (*put)(DB *, DB_TXN *, DBT *, DBT *, uint32_t);
    int (*get)(DB *, DB_TXN *, DBT *, DBT *, uint32_t);
    int (*cursor)(DB *, DB_TXN *, DBC **, uint32_t);
    int (*sync)(DB *, uint32_t);
    void (*err)(DB *, int, const char *, ...);
    /* the stand-in's own state */
    FILE *f;
    uint32_t pagesize;
    uint8_t *page; uint32_t page_used;
    uint8_t *recs; uint64_t recs_len, recs_cap;        /* {u32 size, bytes} back to back: what a cursor walks */
    uint64_t n;
    db_recno_t last_key;
};
struct __dbc {
    int (*c_get)(DBC *, DBT *, DBT *, uint32_t);
    int (*c_close)(DBC *);
    int (*get)(DBC *, DBT *, DBT *, uint32_t);
    int (*close)(DBC *);
    DB *db;
    uint64_t pos;
    db_recno_t key;
};

int db_create(DB **, DB_ENV *, uint32_t);
char *db_strerror(int);
#endif
