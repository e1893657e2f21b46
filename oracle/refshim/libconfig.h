/*
 * libconfig.h -- minimal stand-in for libconfig 1.4.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference's src/config-comp/config-dare.c (compiled unmodified from
 * /root/reference) reads six timing parameters of the `dare_global_config` group
 * through these calls.  oracle/refshim/glue.c implements them with a ~60-line
 * scanner for the `name = value;` syntax of target/nodes.local.cfg.  Only the names
 * config-dare.c uses exist.
 */
#ifndef APUS_FAKE_LIBCONFIG_H
#define APUS_FAKE_LIBCONFIG_H

#define CONFIG_TRUE  1
#define CONFIG_FALSE 0

typedef struct config_setting_t {
    char   name[64];
    char   value[128];
    int    is_group;
    int    n_children;
    struct config_setting_t *children;
} config_setting_t;

typedef struct config_t {
    config_setting_t root;
    const char *error_text;
    const char *error_file;
    int error_line;
} config_t;

void config_init(config_t *config);
void config_destroy(config_t *config);
int  config_read_file(config_t *config, const char *filename);
config_setting_t *config_lookup(const config_t *config, const char *path);
int  config_setting_lookup_float(const config_setting_t *setting, const char *name, double *value);
int  config_setting_lookup_int64(const config_setting_t *setting, const char *name, long long *value);
int  config_lookup_int(const config_t *config, const char *path, int *value);                 /* config-proxy.c */
int  config_lookup_string(const config_t *config, const char *path, const char **value);

#define config_error_text(c) ((c)->error_text ? (c)->error_text : "")
#define config_error_file(c) ((c)->error_file ? (c)->error_file : "")
#define config_error_line(c) ((c)->error_line)

#endif
