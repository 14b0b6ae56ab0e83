/* Minimal declaration shim for the system libsqlite3.so.0 (the image ships the
 * library but not its header).  TEST INFRASTRUCTURE ONLY: lets oracle/Makefile
 * compile the reference's BLAST-DB reader (src/data/blastdb/blastdb.cpp:22,
 * 119-472), which is never executed on .dmnd / FASTA inputs.  Values are the
 * public, stable SQLite C-API constants. */
#ifndef ORACLE_SQLITE3_SHIM_H
#define ORACLE_SQLITE3_SHIM_H
#ifdef __cplusplus
extern "C" {
#endif
typedef struct sqlite3 sqlite3;
typedef struct sqlite3_stmt sqlite3_stmt;
#define SQLITE_OK 0
#define SQLITE_ROW 100
#define SQLITE_DONE 101
#define SQLITE_OPEN_READONLY 0x00000001
int sqlite3_open_v2(const char* filename, sqlite3** db, int flags, const char* vfs);
const char* sqlite3_errmsg(sqlite3*);
int sqlite3_close(sqlite3*);
int sqlite3_prepare_v2(sqlite3* db, const char* sql, int nbyte, sqlite3_stmt** stmt, const char** tail);
int sqlite3_step(sqlite3_stmt*);
int sqlite3_column_int(sqlite3_stmt*, int col);
int sqlite3_finalize(sqlite3_stmt*);
int sqlite3_bind_int(sqlite3_stmt*, int, int);
#ifdef __cplusplus
}
#endif
#endif
