/*
 * glue_store.c -- length of the record BerkeleyDB is handed for one persisted entry, worked out with the
 * reference's OWN structs (src/include/proxy/proxy.h:57-96) the way stablestorage_save_request does
 * (src/proxy/proxy.c:268-291).  TEST INFRASTRUCTURE ONLY; a translation unit of its own because proxy.h
 * and the dare headers cannot be included together (both guard a debug.h with DEBUG_H, <error.h> vs the
 * `error` macro).  `d` = &entry->clt_id (dare_server.c:1802).
 */
#include <stddef.h>
#include <stdint.h>
#include "proxy/proxy.h"

size_t glue_store_record_size(const void *d)
{
    const proxy_msg_header *header = (const proxy_msg_header *)d;
    switch (header->action) {
    case CONNECT: return PROXY_CONNECT_MSG_SIZE;
    case SEND: { const proxy_send_msg *send_msg = (const proxy_send_msg *)d; return PROXY_SEND_MSG_SIZE(send_msg); }
    case CLOSE: return PROXY_CLOSE_MSG_SIZE;
    }
    return 0;
}
