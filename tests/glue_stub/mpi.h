/* tests/glue_stub/mpi.h -- TEST-ONLY declarations of the MPI calls gadget_glue.c makes (syntax check without an MPI installation) */
#ifndef MPI_STUB_H
#define MPI_STUB_H
typedef int MPI_Comm, MPI_Datatype, MPI_Op, MPI_Request;
typedef struct { int s; } MPI_Status;
#define MPI_COMM_WORLD 0
#define MPI_SUCCESS 0
#define MPI_IN_PLACE ((void *)1)
#define MPI_STATUSES_IGNORE ((MPI_Status *)0)
#define MPI_BYTE 1
#define MPI_INT 2
#define MPI_DOUBLE 3
#define MPI_LONG_LONG 4
#define MPI_SUM 1
#define MPI_MIN 2
#define MPI_MAX 3
int MPI_Allreduce(const void *, void *, int, MPI_Datatype, MPI_Op, MPI_Comm);
int MPI_Barrier(MPI_Comm);
int MPI_Allgatherv(const void *, int, MPI_Datatype, void *, const int *, const int *, MPI_Datatype, MPI_Comm);
int MPI_Bcast(void *, int, MPI_Datatype, int, MPI_Comm);
int MPI_Allgather(const void *, int, MPI_Datatype, void *, int, MPI_Datatype, MPI_Comm);
int MPI_Alltoall(const void *, int, MPI_Datatype, void *, int, MPI_Datatype, MPI_Comm);
int MPI_Alltoallv(const void *, const int *, const int *, MPI_Datatype, void *, const int *, const int *, MPI_Datatype, MPI_Comm);
int MPI_Irecv(void *, int, MPI_Datatype, int, int, MPI_Comm, MPI_Request *);
int MPI_Isend(const void *, int, MPI_Datatype, int, int, MPI_Comm, MPI_Request *);
int MPI_Waitall(int, MPI_Request *, MPI_Status *);
#endif
