/*
 * mpm_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * Serial CPU restatement (plain C99) of claymore's single-GPU substep pipeline (Projects/GMPM):
 *     initial_setup -> { update_grid_velocity_query_max -> g2p2g -> partition rebuild }*
 * Every stage follows the reference kernel it names, thread loops replaced by sequential loops in
 * block/thread order (so float sums and bucket orders are one admissible outcome of the reference's
 * atomics).  It is the parity checker for the HIP engine and the "port" CPU baseline of bench.py.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (claymore_amd/, libclaymore_hip.so) never does.
 *
 * Parity pin: the per-particle functions (mpm_oracle_math.h) are checked against golden vectors produced
 * by the reference's own code (tests/golden); the pipeline around them has no reference-run output to
 * compare with (no nvcc / NVIDIA GPU here, and the reference has no tests) and is pinned by invariants
 * (mass / momentum / particle-count conservation, table consistency) in tests/test_oracle_pipeline.py.
 *
 * Exports the C ABI of include/claymore_amd.h with the prefix mpmo_.
 */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "../include/claymore_amd.h"
#include "mpm_oracle_math.h"

#define ORC_BIN 32	   /* settings.h:77 G_BIN_CAPACITY */
#define ORC_BLOCKVOL 64 /* settings.h:70 G_BLOCKVOLUME */
#define ORC_MAX_MODELS 8

typedef struct {
	int* index_table; /* G^3 ints, -1 = empty (hash_table.cuh:107-112) */
	int* keys;		  /* capacity*3 */
	int count;
} orc_partition;

typedef struct {
	float* bins; /* [bin][channel][slot], particle_buffer.cuh:17-41 (channel stride = ORC_BIN floats) */
	size_t bin_cap;
	int* cell_counts;  /* [block][64] */
	int* cellbuckets;  /* [block][64][max_ppc] */
	int* blockbuckets; /* [block][ppb] */
	int* bucket_sizes; /* [block] */
	int* bin_offsets;  /* [block] */
} orc_pbuf;

typedef struct {
	int material;
	mpm_material_params p;
	float mass, volume, mu, lambda, bm;
	int nch;
	size_t n;
	float* xyz; /* particle array (input, and output of retrieve) */
	float v0[3];
	orc_pbuf buf[2];
	int64_t bincount;
} orc_model;

typedef struct mpmo_ctx {
	mpm_config cfg;
	int G;	 /* blocks per axis */
	int ppb; /* particles per block capacity = max_ppc*64 (settings.h:78) */
	float dx, dx_inv, d_inv;
	size_t cap; /* block capacity */
	orc_partition part[2];
	float* grid[2]; /* [block][4][64] */
	int rollid;
	int pbc, nbc, ebc;
	int nmodels;
	orc_model models[ORC_MAX_MODELS];
	int *marks, *sources, *destinations, *bin_sizes;
	int ready;
	/* MGSP halo state (Projects/MGSP/hash_table.cuh:24-69, halo_buffer.cuh) */
	int* overlap;
	int *halo_list, *inner_list;
	int n_halo, n_inner;
	int* send_ids[32];
	int send_count[32];
	int halo_tagged;
	float last_max_vel_sqr;
	int peer_rows_max;
	int threads; /* OpenMP threads over particle blocks in G2P2G (timing only; 1 = the deterministic serial order) */
	/* collision object (Projects/MGSP/boundary_condition.cuh) */
	int has_collision;
	mpm_collision_object col;
	float* sdf4; /* [node][4] = {sdis, gradx, grady, gradz} */
	mpm_timers timers;
	char err[256];
} mpmo_ctx;

static double now_ms(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* ------------------------------------------------------------------------------------------------ */
int mpmo_default_config(int domain_bits, mpm_config* cfg) {
	if(!cfg || domain_bits < 4 || domain_bits > 10) return MPM_ERR_INVALID;
	memset(cfg, 0, sizeof(*cfg));
	cfg->domain_bits	 = domain_bits;
	cfg->max_ppc		 = 128;	  /* settings.h:75 */
	cfg->boundary_blocks = 2;	  /* settings.h:63 */
	cfg->gravity		 = -9.8f; /* settings.h:85 */
	cfg->cfl			 = 0.5f;  /* settings.h:53 */
	cfg->max_blocks		 = 0;
	cfg->grow			 = 1; /* the oracle does not re-allocate: with grow set, max_blocks is only the INITIAL capacity and the
						   oracle sizes itself from the models (6 x particle blocks, as with max_blocks = 0) */
	return MPM_OK;
}

/* oracle-only experiment knob (tools/sand_drift_study.py): converged double-precision SVD instead of the reference's */
int mpmo_set_exact_svd(int on) {
	orc_exact_svd_enabled = on ? 1 : 0;
	return MPM_OK;
}

/* oracle-only: OpenMP threads for G2P2G, the grid update and the order-independent loops of the rebuild (bench.py cpu_baseline); 1 restores
 * the deterministic serial order */
int mpmo_set_threads(mpmo_ctx* c, int n) {
	if(!c || n < 1) return MPM_ERR_INVALID;
	c->threads = n;
	return MPM_OK;
}

int mpmo_default_material(int material, int domain_bits, mpm_material_params* p) {
	if(!p) return MPM_ERR_INVALID;
	memset(p, 0, sizeof(*p));
	const float n	   = (float) (1u << domain_bits);
	p->rho			   = 1e3f;		   /* settings.h:81 */
	p->youngs_modulus  = 5e3f;		   /* settings.h:82 */
	p->poisson_ratio   = 0.4f;		   /* settings.h:83 */
	const float vol1   = 1.0f / n / n / n / 8.0f;
	const float vol10  = 10.f / n / n / n / 8.0f; /* particle_buffer.cuh:176,:203 (sic) */
	switch(material) {
		case MPM_J_FLUID: /* particle_buffer.cuh:148-153 */
			p->volume	 = vol1;
			p->bulk		 = 4e4f;
			p->gamma	 = 7.15f;
			p->viscosity = 0.01f;
			break;
		case MPM_FIXED_COROTATED: p->volume = vol10; break;
		case MPM_SAND: /* particle_buffer.cuh:202-218 */
			p->volume			 = vol10;
			p->cohesion			 = 0.f;
			p->beta				 = 1.f;
			p->yield_surface	 = 0.816496580927726f * 2.f * 0.5f / (3.f - 0.5f);
			p->volume_correction = 1;
			p->log_jp0			 = 0.f;
			break;
		case MPM_NACC: /* particle_buffer.cuh:231-247 */
			p->volume		= vol1;
			p->xi			= 0.8f;
			p->beta			= 0.5f;
			p->msqr			= 3.423772074299613f;
			p->hardening_on = 1;
			p->log_jp0		= -0.01f;
			break;
		default: return MPM_ERR_INVALID;
	}
	return MPM_OK;
}

int mpmo_create(const mpm_config* cfg, int device, mpmo_ctx** out) {
	(void) device;
	if(!cfg || !out) return MPM_ERR_INVALID;
	if(cfg->domain_bits < 4 || cfg->domain_bits > 10) return MPM_ERR_INVALID;
	if(cfg->max_ppc < 1 || cfg->max_ppc > 128 || (cfg->max_ppc & (cfg->max_ppc - 1))) return MPM_ERR_INVALID;
	mpmo_ctx* c = (mpmo_ctx*) calloc(1, sizeof(mpmo_ctx));
	c->cfg		= *cfg;
	c->G		= 1 << (cfg->domain_bits - 2);
	c->ppb		= cfg->max_ppc * ORC_BLOCKVOL;
	c->dx_inv	= (float) (1 << cfg->domain_bits); /* settings.h:60 */
	c->dx		= 1.f / c->dx_inv;				   /* settings.h:64 */
	c->d_inv	= 4.f * c->dx_inv * c->dx_inv;	   /* settings.h:66 */
	*out		= c;
	return MPM_OK;
}

static void free_pbuf(orc_pbuf* b) {
	free(b->bins);
	free(b->cell_counts);
	free(b->cellbuckets);
	free(b->blockbuckets);
	free(b->bucket_sizes);
	free(b->bin_offsets);
}

void mpmo_destroy(mpmo_ctx* c) {
	if(!c) return;
	for(int i = 0; i < 2; ++i) {
		free(c->part[i].index_table);
		free(c->part[i].keys);
		free(c->grid[i]);
	}
	for(int m = 0; m < c->nmodels; ++m) {
		free(c->models[m].xyz);
		free_pbuf(&c->models[m].buf[0]);
		free_pbuf(&c->models[m].buf[1]);
	}
	free(c->sdf4);
	free(c->marks);
	free(c->sources);
	free(c->destinations);
	free(c->bin_sizes);
	free(c);
}

const char* mpmo_last_error(const mpmo_ctx* c) {
	return c ? c->err : "null context";
}

static int fail(mpmo_ctx* c, int code, const char* msg) {
	snprintf(c->err, sizeof(c->err), "%s", msg);
	return code;
}

int mpmo_add_model(mpmo_ctx* c, int material, const mpm_material_params* p, const float* xyz, size_t n, const float v0[3], int* model_id) {
	if(!c || !p || !xyz || c->ready) return MPM_ERR_INVALID;
	if(c->nmodels >= ORC_MAX_MODELS) return fail(c, MPM_ERR_CAPACITY, "too many models");
	orc_model* m = &c->models[c->nmodels];
	memset(m, 0, sizeof(*m));
	m->material = material;
	m->p		= *p;
	m->volume	= p->volume;
	m->mass		= p->volume * p->rho; /* particle_buffer.cuh:158,:186,:252 */
	const float e = p->youngs_modulus, nu = p->poisson_ratio;
	m->lambda = e * nu / ((1 + nu) * (1 - 2 * nu)); /* particle_buffer.cuh:187 */
	m->mu	  = e / (2 * (1 + nu));					/* :188 */
	m->bm	  = 2.f / 3.f * (e / (2 * (1 + nu))) + (e * nu / ((1 + nu) * (1 - 2 * nu))); /* :255 */
	m->nch	  = material == MPM_J_FLUID ? 4 : (material == MPM_FIXED_COROTATED ? 12 : 13);
	m->n	  = n;
	m->xyz	  = (float*) malloc(sizeof(float) * 3 * (n ? n : 1));
	memcpy(m->xyz, xyz, sizeof(float) * 3 * n);
	for(int d = 0; d < 3; ++d) m->v0[d] = v0 ? v0[d] : 0.f;
	if(model_id) *model_id = c->nmodels;
	c->nmodels++;
	return MPM_OK;
}

/* ---- Partition (Projects/GMPM/hash_table.cuh:76-135) ---- */
static inline int key_in_range(const mpmo_ctx* c, int x, int y, int z) {
	return x >= 0 && y >= 0 && z >= 0 && x < c->G && y < c->G && z < c->G;
}
static inline size_t key_index(const mpmo_ctx* c, int x, int y, int z) {
	return ((size_t) x * c->G + y) * c->G + z; /* CompactDomain row-major, StructuralDeclaration.h:235-251 */
}
static inline int part_query(const mpmo_ctx* c, const orc_partition* p, int x, int y, int z) {
	if(!key_in_range(c, x, y, z)) return -1; /* reference: out-of-range is undefined behaviour */
	return p->index_table[key_index(c, x, y, z)];
}
/* hash_table.cuh:118-127 insert */
static inline int part_insert(mpmo_ctx* c, orc_partition* p, int x, int y, int z) {
	if(!key_in_range(c, x, y, z)) return -1;
	size_t i = key_index(c, x, y, z);
	if(p->index_table[i] != -1) return -1;
	if((size_t) p->count >= c->cap) {
		p->count++; /* keep counting so that the caller can report the overflow */
		return -1;
	}
	int idx				 = p->count++;
	p->index_table[i]	 = idx;
	p->keys[3 * idx + 0] = x;
	p->keys[3 * idx + 1] = y;
	p->keys[3 * idx + 2] = z;
	return idx;
}
/* bench.py's multi-core cpu_baseline (mpmo_set_threads(n > 1)): besides G2P2G the loops below run on the OpenMP threads too - those whose
 * iterations are independent and whose result does not depend on the order they run in (fills, per-block copies, a max).  The serial
 * inserts that define the block numbering (register_*_blocks, update_partition) stay serial.  With one thread nothing changes. */
#define ORC_PAR _Pragma("omp parallel for schedule(static) num_threads(c->threads > 1 ? c->threads : 1) if(c->threads > 1)")
static void par_fill(const mpmo_ctx* c, void* dst, int byte, size_t bytes) {
	const size_t chunk = (size_t) 1 << 22;
	const long long nchunks = (long long) ((bytes + chunk - 1) / chunk);
	ORC_PAR
	for(long long i = 0; i < nchunks; ++i) {
		const size_t off = (size_t) i * chunk;
		memset((char*) dst + off, byte, bytes - off < chunk ? bytes - off : chunk);
	}
}
static void part_reset_table(mpmo_ctx* c, orc_partition* p) {
	par_fill(c, p->index_table, 0xff, sizeof(int) * (size_t) c->G * c->G * c->G);
}

static inline float* grid_block(float* grid, int blockno) {
	return grid + (size_t) blockno * 256; /* grid_buffer.cuh:12-14: 4 channels x 64 cells */
}
static inline float* bin_ptr(const orc_model* m, const orc_pbuf* b, int binno) {
	return b->bins + (size_t) binno * m->nch * ORC_BIN;
}

/* mgmpm_kernels.cuh:106-115 clear_grid */
static void clear_grid(const mpmo_ctx* c, float* grid, int nblocks) {
	par_fill(c, grid, 0, sizeof(float) * 256 * (size_t) nblocks);
}

/* mgmpm_kernels.cuh:70-84 cell_bucket_to_block: round k takes the k-th particle of every cell, cells ascending */
static void cell_bucket_to_block(mpmo_ctx* c, orc_pbuf* b, int nblocks) {
	const int mp = c->cfg.max_ppc;
	ORC_PAR
	for(int blk = 0; blk < nblocks; ++blk) {
		const int* counts = b->cell_counts + (size_t) blk * ORC_BLOCKVOL;
		int* out		  = b->blockbuckets + (size_t) blk * c->ppb;
		int size		  = 0;
		for(int k = 0; k < mp; ++k) {
			for(int cell = 0; cell < ORC_BLOCKVOL; ++cell) {
				if(k < counts[cell]) out[size++] = b->cellbuckets[(size_t) blk * c->ppb + (size_t) cell * mp + k];
			}
		}
		b->bucket_sizes[blk] = size;
	}
}

/* compute_bin_capacity + exclusive scan (mgmpm_kernels.cuh:86-94, gmpm_simulator.cuh:496-503) over pbc+1 entries */
static int64_t bin_offsets_scan(mpmo_ctx* c, orc_pbuf* b, int pbc) {
	int64_t acc = 0;
	for(int blk = 0; blk <= pbc; ++blk) {
		b->bin_offsets[blk] = (int) acc;
		if(blk < pbc) acc += (b->bucket_sizes[blk] + ORC_BIN - 1) / ORC_BIN;
	}
	(void) c;
	return acc;
}

/* mgmpm_kernels.cuh:117-133 / :135-151 */
static void register_neighbor_blocks(mpmo_ctx* c, orc_partition* p, int pbc) {
	for(int b = 0; b < pbc; ++b) {
		const int x = p->keys[3 * b], y = p->keys[3 * b + 1], z = p->keys[3 * b + 2];
		for(int i = 0; i < 2; ++i)
			for(int j = 0; j < 2; ++j)
				for(int k = 0; k < 2; ++k) part_insert(c, p, x + i, y + j, z + k);
	}
}
static void register_exterior_blocks(mpmo_ctx* c, orc_partition* p, int pbc) {
	for(int b = 0; b < pbc; ++b) {
		const int x = p->keys[3 * b], y = p->keys[3 * b + 1], z = p->keys[3 * b + 2];
		for(int i = -1; i < 2; ++i)
			for(int j = -1; j < 2; ++j)
				for(int k = -1; k < 2; ++k) part_insert(c, p, x + i, y + j, z + k);
	}
}

static void alloc_pbuf(mpmo_ctx* c, orc_model* m, orc_pbuf* b, size_t bin_cap) {
	b->bin_cap		= bin_cap;
	b->bins			= (float*) calloc(bin_cap * m->nch * ORC_BIN, sizeof(float));
	b->cell_counts	= (int*) calloc(c->cap * ORC_BLOCKVOL, sizeof(int));
	b->cellbuckets	= (int*) calloc(c->cap * c->ppb, sizeof(int));
	b->blockbuckets = (int*) calloc(c->cap * c->ppb, sizeof(int));
	b->bucket_sizes = (int*) calloc(c->cap + 1, sizeof(int));
	b->bin_offsets	= (int*) calloc(c->cap + 1, sizeof(int));
}

/* ParticleBufferImpl::add_advection, Projects/GMPM/particle_buffer.cuh:100-135 */
static inline void add_advection(mpmo_ctx* c, orc_pbuf* b, const orc_partition* table, int cx, int cy, int cz, int dirtag, int pidib) {
	const int bx = cx / 4, by = cy / 4, bz = cz / 4; /* C++ truncating ivec3 division, Vec.h */
	const int blockno = part_query(c, table, bx, by, bz);
	if(blockno == -1) return; /* particle is lost, :105-113 */
	const int cellno = ((cx & 3) << 4) | ((cy & 3) << 2) | (cz & 3);
	int* cnt = b->cell_counts + (size_t) blockno * ORC_BLOCKVOL + cellno;
	int slot;
#pragma omp atomic capture
	slot = (*cnt)++;
	if(slot >= c->cfg.max_ppc) {
#pragma omp atomic
		(*cnt)--;
		return;
	}
	b->cellbuckets[(size_t) blockno * c->ppb + (size_t) cellno * c->cfg.max_ppc + slot] = (dirtag * c->ppb) | pidib;
}

/* ---- initial_setup, Projects/GMPM/gmpm_simulator.cuh:637-781 ---- */
int mpmo_initial_setup(mpmo_ctx* c) {
	if(!c || c->ready || c->nmodels == 0) return MPM_ERR_INVALID;
	const int r = c->rollid, n = r ^ 1;
	/* capacity: count distinct particle blocks first (the reference uses the compile-time G_MAX_ACTIVE_BLOCK) */
	size_t table = (size_t) c->G * c->G * c->G;
	if(c->cfg.max_blocks > 0 && !c->cfg.grow) {
		c->cap = (size_t) c->cfg.max_blocks;
	} else {
		unsigned char* seen = (unsigned char*) calloc(table, 1);
		size_t pb			= 0;
		for(int mi = 0; mi < c->nmodels; ++mi) {
			orc_model* m = &c->models[mi];
			for(size_t i = 0; i < m->n; ++i) {
				int k[3];
				for(int d = 0; d < 3; ++d) k[d] = (orc_node_index(m->xyz[3 * i + d], c->dx_inv) - 2) / 4;
				if(!key_in_range(c, k[0], k[1], k[2])) continue;
				size_t idx = key_index(c, k[0], k[1], k[2]);
				if(!seen[idx]) {
					seen[idx] = 1;
					pb++;
				}
			}
		}
		free(seen);
		c->cap = pb * 6 + 4096; /* room for the {-1,0,1}^3 ring and for growth */
		if(c->cap > table) c->cap = table;
	}
	for(int i = 0; i < 2; ++i) {
		c->part[i].index_table = (int*) malloc(sizeof(int) * table);
		c->part[i].keys		   = (int*) calloc(c->cap * 3, sizeof(int));
		c->part[i].count	   = 0;
		part_reset_table(c, &c->part[i]);
		c->grid[i] = (float*) calloc(c->cap * 256, sizeof(float));
	}
	c->marks		= (int*) calloc(c->cap + 1, sizeof(int));
	c->sources		= (int*) calloc(c->cap + 1, sizeof(int));
	c->destinations = (int*) calloc(c->cap + 1, sizeof(int));
	c->bin_sizes	= (int*) calloc(c->cap + 1, sizeof(int));
	for(int mi = 0; mi < c->nmodels; ++mi) {
		orc_model* m   = &c->models[mi];
		size_t bin_cap = m->n / ORC_BIN + c->cap; /* gmpm_simulator.cuh:174 */
		alloc_pbuf(c, m, &m->buf[0], bin_cap);
		alloc_pbuf(c, m, &m->buf[1], bin_cap);
	}

	orc_partition* P = &c->part[n];
	/* activate_blocks, mgmpm_kernels.cuh:21-34 */
	for(int mi = 0; mi < c->nmodels; ++mi) {
		orc_model* m = &c->models[mi];
		for(size_t i = 0; i < m->n; ++i) {
			int k[3];
			for(int d = 0; d < 3; ++d) k[d] = (orc_node_index(m->xyz[3 * i + d], c->dx_inv) - 2) / 4;
			part_insert(c, P, k[0], k[1], k[2]);
		}
	}
	c->pbc = P->count;
	if((size_t) c->pbc > c->cap) return fail(c, MPM_ERR_CAPACITY, "Too much active blocks");
	/* build_particle_cell_buckets, mgmpm_kernels.cuh:36-68 (into bins[rollid]) */
	for(int mi = 0; mi < c->nmodels; ++mi) {
		orc_model* m = &c->models[mi];
		orc_pbuf* b	 = &m->buf[r];
		for(size_t i = 0; i < m->n; ++i) {
			int co[3];
			for(int d = 0; d < 3; ++d) co[d] = orc_node_index(m->xyz[3 * i + d], c->dx_inv) - 2;
			int blockno = part_query(c, P, co[0] / 4, co[1] / 4, co[2] / 4);
			if(blockno < 0) continue;
			int cellno = (co[0] & 3) * 16 + (co[1] & 3) * 4 + (co[2] & 3);
			int* cnt   = b->cell_counts + (size_t) blockno * ORC_BLOCKVOL + cellno;
			int slot   = (*cnt)++;
			if(slot >= c->cfg.max_ppc) {
				(*cnt)--;
				continue;
			}
			b->cellbuckets[(size_t) blockno * c->ppb + (size_t) cellno * c->cfg.max_ppc + slot] = (int) i;
		}
		/* cell_bucket_to_block + bin offsets + array_to_buffer (gmpm_simulator.cuh:676-704) */
		cell_bucket_to_block(c, b, c->pbc);
		b->bucket_sizes[c->pbc] = 0;
		m->bincount				= bin_offsets_scan(c, b, c->pbc);
		if((size_t) m->bincount > b->bin_cap) return fail(c, MPM_ERR_CAPACITY, "bin capacity");
		/* array_to_buffer, mgmpm_kernels.cuh:221-323 */
		for(int blk = 0; blk < c->pbc; ++blk) {
			const int cnt	  = b->bucket_sizes[blk];
			const int* bucket = b->blockbuckets + (size_t) blk * c->ppb;
			for(int pidib = 0; pidib < cnt; ++pidib) {
				const int pid = bucket[pidib];
				float* bin	  = bin_ptr(m, b, b->bin_offsets[blk] + pidib / ORC_BIN);
				const int s	  = pidib % ORC_BIN;
				bin[0 * ORC_BIN + s] = m->xyz[3 * pid + 0];
				bin[1 * ORC_BIN + s] = m->xyz[3 * pid + 1];
				bin[2 * ORC_BIN + s] = m->xyz[3 * pid + 2];
				if(m->material == MPM_J_FLUID) {
					bin[3 * ORC_BIN + s] = 1.0f;
				} else {
					for(int d = 0; d < 9; ++d) bin[(3 + d) * ORC_BIN + s] = (d % 4 == 0) ? 1.f : 0.f;
					if(m->nch == 13) bin[12 * ORC_BIN + s] = m->p.log_jp0;
				}
			}
		}
	}
	/* register neighbors / exterior (gmpm_simulator.cuh:706-734) */
	register_neighbor_blocks(c, P, c->pbc);
	c->nbc = P->count;
	if((size_t) c->nbc > c->cap) return fail(c, MPM_ERR_CAPACITY, "Too much neighbour blocks");
	register_exterior_blocks(c, P, c->pbc);
	c->ebc = P->count;
	if((size_t) c->ebc > c->cap) return fail(c, MPM_ERR_CAPACITY, "Too much exterior blocks");
	/* copy partition + bucket metadata to the other roll (gmpm_simulator.cuh:745-759) */
	memcpy(c->part[r].index_table, P->index_table, sizeof(int) * table);
	memcpy(c->part[r].keys, P->keys, sizeof(int) * 3 * (size_t) c->ebc);
	c->part[r].count = P->count;
	for(int mi = 0; mi < c->nmodels; ++mi) {
		orc_model* m = &c->models[mi];
		memcpy(m->buf[n].bin_offsets, m->buf[r].bin_offsets, sizeof(int) * ((size_t) c->pbc + 1));
		memcpy(m->buf[n].bucket_sizes, m->buf[r].bucket_sizes, sizeof(int) * (size_t) c->pbc);
	}
	/* rasterize, mgmpm_kernels.cuh:153-219, and init_adv_bucket :96-104 */
	clear_grid(c, c->grid[0], c->nbc);
	for(int mi = 0; mi < c->nmodels; ++mi) {
		orc_model* m = &c->models[mi];
		for(size_t i = 0; i < m->n; ++i) {
			const float* pos = m->xyz + 3 * i;
			int base[3];
			float local[3], dws[3][3];
			for(int d = 0; d < 3; ++d) {
				base[d]	 = orc_node_index(pos[d], c->dx_inv) - 1;
				local[d] = pos[d] - base[d] * c->dx;
				orc_bspline_weight(local[d], c->dx_inv, dws[d]);
			}
			for(int ii = 0; ii < 3; ++ii)
				for(int jj = 0; jj < 3; ++jj)
					for(int kk = 0; kk < 3; ++kk) {
						const int gx = base[0] + ii, gy = base[1] + jj, gz = base[2] + kk;
						const float w  = dws[0][ii] * dws[1][jj] * dws[2][kk];
						const float wm = m->mass * w;
						const int bno  = part_query(c, &c->part[r], gx >> 2, gy >> 2, gz >> 2);
						if(bno < 0) continue;
						float* g		= grid_block(c->grid[0], bno);
						const int cell	= (gx & 3) * 16 + (gy & 3) * 4 + (gz & 3);
						g[cell] += wm;
						g[64 + cell] += wm * m->v0[0];
						g[128 + cell] += wm * m->v0[1];
						g[192 + cell] += wm * m->v0[2];
					}
		}
		orc_pbuf* bn = &m->buf[n];
		for(int blk = 0; blk < c->pbc; ++blk) {
			int* bucket = bn->blockbuckets + (size_t) blk * c->ppb;
			for(int pidib = 0; pidib < bn->bucket_sizes[blk]; ++pidib) bucket[pidib] = (orc_dir_offset(0, 0, 0) * c->ppb) | pidib;
		}
	}
	c->ready = 1;
	return MPM_OK;
}

/* ---- update_grid_velocity_query_max, Projects/GMPM/mgmpm_kernels.cuh:325-420 ---- */
/* ---- SignedDistanceGrid, Projects/MGSP/boundary_condition.cuh:25-250 ---- */
int mpmo_default_collision_object(mpm_collision_object* o) {
	if(!o) return MPM_ERR_INVALID;
	memset(o, 0, sizeof(*o));
	o->type		= MPM_BOUNDARY_STICKY; /* :46 */
	o->friction = 0.3f;				   /* :45 */
	o->scale	= 1.0f;				   /* :44 */
	o->dsdt		= 0.0f;				   /* :43 */
	o->rot_mat[0] = o->rot_mat[4] = o->rot_mat[8] = 1.f; /* :47-48 */
	return MPM_OK;
}

int mpmo_set_collision_object(mpmo_ctx* c, const mpm_collision_object* obj, const float* sdf, const float* gx, const float* gy, const float* gz) {
	if(!c) return MPM_ERR_INVALID;
	free(c->sdf4);
	c->sdf4			 = NULL;
	c->has_collision = 0;
	if(!obj) return MPM_OK;
	if(!sdf || !gx || !gy || !gz) return MPM_ERR_INVALID;
	const size_t N = (size_t) c->G * 4, n = N * N * N;
	c->sdf4 = (float*) malloc(sizeof(float) * 4 * n);
	if(!c->sdf4) return MPM_ERR_DEVICE;
	for(size_t i = 0; i < n; ++i) {
		c->sdf4[4 * i]	   = sdf[i];
		c->sdf4[4 * i + 1] = gx[i];
		c->sdf4[4 * i + 2] = gy[i];
		c->sdf4[4 * i + 3] = gz[i];
	}
	c->col			 = *obj;
	c->has_collision = 1;
	return MPM_OK;
}

/* rot_angle_to_matrix (:68-91): element (i, j) of the reference's vec3x3 lives at [3 i + j] */
static void col_rot_angle_to_matrix(float omega, int dim, float* res) {
	for(int i = 0; i < 9; ++i) res[i] = 0.f;
	if(dim == 0) {
		res[0] = 1;
		res[4] = res[8] = cosf(omega);
		res[7] = res[5] = sinf(omega);
		res[5]			= -res[5];
	} else if(dim == 1) {
		res[4] = 1;
		res[0] = res[8] = cosf(omega);
		res[6] = res[2] = sinf(omega);
		res[6]			= -res[6];
	} else {
		res[8] = 1;
		res[0] = res[4] = cosf(omega);
		res[3] = res[1] = sinf(omega);
		res[1]			= -res[1];
	}
}
/* vec_cross_mul_vec_3d, MatrixUtils.h:53-58 and vec3_cross_vec3, boundary_condition.cuh:93-96 (both with plus signs) */
static void col_cross(float* out, const float* a, const float* b) {
	out[0] = a[1] * b[2] + a[2] * b[1];
	out[1] = a[2] * b[0] + a[0] * b[2];
	out[2] = a[0] * b[1] + a[1] * b[0];
}
/* get_signed_distance_and_normal (:99-140) */
static float col_signed_distance_and_normal(const mpmo_ctx* c, const float* x, float* normal) {
	const float dx = c->dx;
	const size_t N = (size_t) c->G * 4;
	int g_cid[3];
	for(int d = 0; d < 3; ++d) g_cid[d] = (int) (x[d] / dx);
	float sdis = 0.f;
	normal[0] = normal[1] = normal[2] = 0.f;
	float w1[3][2];
	for(int d = 0; d < 3; ++d) {
		const float dis_lb = x[d] - ((float) g_cid[d] * dx);
		w1[d][0]		   = 1 - dis_lb / dx;
		w1[d][1]		   = dis_lb / dx;
	}
	for(int i = 0; i < 2; ++i)
		for(int j = 0; j < 2; ++j)
			for(int k = 0; k < 2; ++k) {
				const float w  = w1[0][i] * w1[1][j] * w1[2][k];
				const float* v = c->sdf4 + 4 * ((((size_t) (g_cid[0] + i)) * N + (size_t) (g_cid[1] + j)) * N + (size_t) (g_cid[2] + k));
				sdis += w * v[0];
				normal[0] += w * v[1];
				normal[1] += w * v[2];
				normal[2] += w * v[3];
			}
	const float nn = sqrtf(normal[0] * normal[0] + normal[1] * normal[1] + normal[2] * normal[2]);
	for(int d = 0; d < 3; ++d) normal[d] /= nn;
	return sdis;
}
/* query_sdf (:141-146) */
static int col_query_sdf(const mpmo_ctx* c, float* normal, const float* x) {
	const float lo = (float) c->cfg.boundary_blocks * c->dx * 4.f;
	const float hi = (float) (c->G - c->cfg.boundary_blocks) * 4.f * c->dx;
	if(x[0] < lo || x[0] >= hi || x[1] < lo || x[1] >= hi || x[2] < lo || x[2] >= hi) return 0;
	return col_signed_distance_and_normal(c, x, normal) <= 0.f;
}
/* detect_and_resolve_collision (:164-248) */
static void col_detect_and_resolve(const mpmo_ctx* c, const int* block_id, const int* cell_id, float t, float* vel) {
	const mpm_collision_object* o = &c->col;
	float xmt[3], x[3], x0[3];
	for(int d = 0; d < 3; ++d) xmt[d] = (float) (block_id[d] * 4 + cell_id[d]) * c->dx - (o->trans[d] + o->trans_vel[d] * t);
	float rot[9], tmp[9], prev[9];
	memcpy(rot, o->rot_mat, sizeof(rot));
	{
		const float inv = 1.f / (1.f + o->dsdt * t);
		for(int d = 0; d < 3; ++d) x0[d] = xmt[d] * inv;
		for(int dim = 0; dim < 3; ++dim) {
			col_rot_angle_to_matrix(o->omega[dim] * t, dim, tmp);
			memcpy(prev, rot, sizeof(rot));
			orc_matmul3(prev, tmp, rot); /* matrix_matrix_multiplication_3d on the raw arrays (:177-187) */
		}
		/* mat_t_mul_vec_3d, MatrixUtils.h:45-49 */
		x[0] = rot[0] * x0[0] + rot[1] * x0[1] + rot[2] * x0[2];
		x[1] = rot[3] * x0[0] + rot[4] * x0[1] + rot[5] * x0[2];
		x[2] = rot[6] * x0[0] + rot[7] * x0[1] + rot[8] * x0[2];
	}
	for(int d = 0; d < 3; ++d) x[d] = x[d] * o->scale + o->trans[d];
	float n[3];
	if(!col_query_sdf(c, n, x)) return;
	/* object velocity in deformation space (:197-204) */
	float v_obj[3], radius[3], mat_vel[3], rot_v[3];
	col_cross(v_obj, o->omega, xmt);
	for(int d = 0; d < 3; ++d) v_obj[d] += xmt[d] * (o->dsdt / o->scale);
	for(int d = 0; d < 3; ++d) radius[d] = x[d] - o->trans[d];
	col_cross(mat_vel, o->omega, radius); /* get_material_velocity (:59-65) */
	for(int d = 0; d < 3; ++d) mat_vel[d] += o->trans_vel[d];
	/* matrix_vector_multiplication_3d, MatrixUtils.h:210-214 */
	rot_v[0] = rot[0] * mat_vel[0] + rot[3] * mat_vel[1] + rot[6] * mat_vel[2];
	rot_v[1] = rot[1] * mat_vel[0] + rot[4] * mat_vel[1] + rot[7] * mat_vel[2];
	rot_v[2] = rot[2] * mat_vel[0] + rot[5] * mat_vel[1] + rot[8] * mat_vel[2];
	for(int d = 0; d < 3; ++d) v_obj[d] += rot_v[d] * o->scale + o->trans_vel[d];
	for(int d = 0; d < 3; ++d) vel[d] -= v_obj[d];
	if(o->type == MPM_BOUNDARY_STICKY) {
		vel[0] = vel[1] = vel[2] = 0.f;
	} else {
		if(o->type == MPM_BOUNDARY_SEPARATE && n[0] == 0.0f && n[1] == 0.0f && n[2] == 0.0f) {
			vel[0] = vel[1] = vel[2] = 0.f;
			return; /* (:227-230): returns WITHOUT adding the object velocity back */
		}
		float nr[3];
		nr[0] = rot[0] * n[0] + rot[3] * n[1] + rot[6] * n[2];
		nr[1] = rot[1] * n[0] + rot[4] * n[1] + rot[7] * n[2];
		nr[2] = rot[2] * n[0] + rot[5] * n[1] + rot[8] * n[2];
		const float v_dot_n = nr[0] * vel[0] + nr[1] * vel[1] + nr[2] * vel[2];
		if(o->type == MPM_BOUNDARY_SLIP) {
			for(int d = 0; d < 3; ++d) vel[d] -= nr[d] * v_dot_n;
			if(o->friction > 0.0f && v_dot_n < 0) {
				const float vel_norm = sqrtf(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]);
				if(-v_dot_n * o->friction < vel_norm) {
					for(int d = 0; d < 3; ++d) vel[d] += vel[d] / vel_norm * (v_dot_n * o->friction);
				} else {
					vel[0] = vel[1] = vel[2] = 0.f;
				}
			}
		} else if(v_dot_n < 0) { /* SEPARATE (:236-247) */
			for(int d = 0; d < 3; ++d) vel[d] -= nr[d] * v_dot_n;
			if(o->friction != 0) {
				const float vel_norm = sqrtf(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]);
				if(-v_dot_n * o->friction < vel_norm) {
					for(int d = 0; d < 3; ++d) vel[d] += vel[d] / vel_norm * (v_dot_n * o->friction);
				} else {
					vel[0] = vel[1] = vel[2] = 0.f;
				}
			}
		}
	}
	for(int d = 0; d < 3; ++d) vel[d] += v_obj[d];
}

/* the cell arithmetic of update_grid_velocity_query_max (mgmpm_kernels.cuh:353-388; with a collision object the MGSP project's overload,
 * Projects/MGSP/mgmpm_kernels.cuh:362-373): one function for the pipeline and for the statement-level pin tests/golden/g19_* */
static inline float orc_grid_cell(const mpmo_ctx* col, const int* key, int cell, float* g, int wx, int wy, int wz, float gravity, float dt) {
	const float mass = g[cell];
	float vel_sqr	 = 0.f;
	if(mass > 0.0f) {
		const float mass_inv = 1.f / mass;
		float v0 = g[64 + cell], v1 = g[128 + cell], v2 = g[192 + cell];
		v0 = wx ? 0.0f : v0 * mass_inv;
		v1 = wy ? 0.0f : v1 * mass_inv;
		v1 += gravity * dt;
		v2 = wz ? 0.0f : v2 * mass_inv;
		if(col) { /* boundary overload, Projects/MGSP/mgmpm_kernels.cuh:362-373 */
			float vel[3]		 = {v0, v1, v2};
			const int cellid[3] = {(cell & 0x30) >> 4, (cell & 0xc) >> 2, cell & 0x3};
			col_detect_and_resolve(col, key, cellid, col->col.time, vel);
			v0 = vel[0];
			v1 = vel[1];
			v2 = vel[2];
			vel_sqr = v0 * v0 + v1 * v1 + v2 * v2; /* vel.dot(vel), then added again below: the reference's 2 |v|^2 */
		}
		g[64 + cell]  = v0;
		g[128 + cell] = v1;
		g[192 + cell] = v2;
		vel_sqr += v0 * v0;
		vel_sqr += v1 * v1;
		vel_sqr += v2 * v2;
	}
	if(isnan(vel_sqr)) vel_sqr = INFINITY;
	return vel_sqr;
}

int mpmo_grid_update(mpmo_ctx* c, float dt, float* max_vel_sqr) {
	if(!c || !c->ready) return MPM_ERR_NOT_READY;
	double t0				= now_ms();
	const orc_partition* P	= &c->part[c->rollid];
	const int bc			= c->cfg.boundary_blocks;
	float maxv				= 0.f;
#pragma omp parallel for schedule(static) reduction(max : maxv) num_threads(c->threads > 1 ? c->threads : 1) if(c->threads > 1)
	for(int b = 0; b < c->nbc; ++b) {
		const int* key = P->keys + 3 * b;
		const int wx   = key[0] < bc || key[0] >= c->G - bc;
		const int wy   = key[1] < bc || key[1] >= c->G - bc;
		const int wz   = key[2] < bc || key[2] >= c->G - bc;
		float* g	   = grid_block(c->grid[0], b);
		for(int cell = 0; cell < 64; ++cell) {
			const float vel_sqr = orc_grid_cell(c->has_collision ? c : NULL, key, cell, g, wx, wy, wz, c->cfg.gravity, dt);
			if(vel_sqr > maxv) maxv = vel_sqr;
		}
	}
	if(max_vel_sqr) *max_vel_sqr = maxv;
	c->timers.grid_update_ms = (float) (now_ms() - t0);
	return MPM_OK;
}

float mpmo_compute_dt(const mpmo_ctx* c, float max_vel, float cur_time, float next_time, float dt_default) {
	return orc_compute_dt(max_vel, cur_time, next_time, dt_default, c->dx, c->cfg.cfl);
}

/* ---- the per-particle body of g2p2g, Projects/GMPM/mgmpm_kernels.cuh:774-905 (with the per-material bodies :470-663) ----
 * One function for the pipeline (g2p2g_model) and for the statement-level pin: tests/golden/g16_* were produced by the reference's own
 * statements of these lines, cut out of its kernel as text (tests/golden/gen/gen_golden_kernel.sh), one particle at a time. */
typedef struct {
	int base[3], arena[3]; /* base_index, its image in the 8^3 arena (:774-797) */
	float vel[3], A[9], pos[3];
	float J, F[9], log_jp; /* what the material body stores */
	float stress[9];	   /* contrib as compute_stress / the J-fluid block leaves it */
	float contrib[9];	   /* after :850 */
	int adv_cell[3], dirtag; /* the arguments of add_advection (:863) */
	int narena[3], discarded;
	float local_pos[3];
} orc_particle_out;
static inline void orc_particle_body(int material, float dx, float dx_inv, float d_inv, float mass, float volume, float mu, float lambda, float bm, const mpm_material_params* prm, const float (*g2p)[8][8][8], float (*p2g)[8][8][8],
									 const float* pos_in, float J, const float* Fold, float log_jp, float dt, float new_dt, orc_particle_out* o) {
	float pos[3] = {pos_in[0], pos_in[1], pos_in[2]};
	memset(o, 0, sizeof(*o));
	o->discarded = 1;
	/* stencil base, weights (:774-797) */
	int base_index[3], arena[3];
	float local_pos[3], dws[3][3];
	for(int d = 0; d < 3; ++d) {
		base_index[d] = orc_node_index(pos[d], dx_inv) - 1;
		local_pos[d]  = pos[d] - base_index[d] * dx;
		orc_bspline_weight(local_pos[d], dx_inv, dws[d]);
		arena[d] = ((base_index[d] - 1) & 3) + 1;
	}
	/* G2P gather (:803-835) */
	float vel[3] = {0.f, 0.f, 0.f};
	float A[9]	 = {0};
	for(int i = 0; i < 3; i++)
		for(int j = 0; j < 3; j++)
			for(int k = 0; k < 3; k++) {
				const float xixp[3] = {(float) i * dx - local_pos[0], (float) j * dx - local_pos[1], (float) k * dx - local_pos[2]};
				const float W		= dws[0][i] * dws[1][j] * dws[2][k];
				const float vi[3]	= {g2p[0][arena[0] + i][arena[1] + j][arena[2] + k], g2p[1][arena[0] + i][arena[1] + j][arena[2] + k], g2p[2][arena[0] + i][arena[1] + j][arena[2] + k]};
				vel[0] += W * vi[0];
				vel[1] += W * vi[1];
				vel[2] += W * vi[2];
				A[0] += W * vi[0] * xixp[0];
				A[1] += W * vi[1] * xixp[0];
				A[2] += W * vi[2] * xixp[0];
				A[3] += W * vi[0] * xixp[1];
				A[4] += W * vi[1] * xixp[1];
				A[5] += W * vi[2] * xixp[1];
				A[6] += W * vi[0] * xixp[2];
				A[7] += W * vi[1] * xixp[2];
				A[8] += W * vi[2] * xixp[2];
			}
	/* advect (:838) */
	for(int d = 0; d < 3; ++d) pos[d] += vel[d] * dt;
	for(int d = 0; d < 3; ++d) {
		o->base[d]	= base_index[d];
		o->arena[d] = arena[d];
		o->vel[d]	= vel[d];
		o->pos[d]	= pos[d];
	}
	for(int d = 0; d < 9; ++d) o->A[d] = A[d];
	/* material update (calculate_contribution_and_store_particle_data, :470-663) */
	float contrib[9];
	o->J	  = J;
	o->log_jp = log_jp;
	if(material == MPM_J_FLUID) {
		o->J = orc_jfluid(J, A, dt, d_inv, volume, prm->bulk, prm->gamma, prm->viscosity, contrib);
	} else {
		float dws9[9], F[9];
		for(int d = 0; d < 9; ++d) dws9[d] = A[d] * dt * d_inv + ((d & 0x3) != 0 ? 0.f : 1.f);
		orc_matmul3(dws9, Fold, F);
		if(material == MPM_FIXED_COROTATED) {
			orc_stress_fixed_corotated(volume, mu, lambda, F, contrib);
		} else if(material == MPM_SAND) {
			orc_stress_sand(volume, mu, lambda, prm->cohesion, prm->beta, prm->yield_surface, prm->volume_correction, F, &log_jp, contrib);
		} else {
			orc_stress_nacc(volume, mu, lambda, bm, prm->xi, prm->beta, prm->msqr, prm->hardening_on, F, &log_jp, contrib);
		}
		for(int d = 0; d < 9; ++d) o->F[d] = F[d];
		o->log_jp = log_jp;
	}
	for(int d = 0; d < 9; ++d) o->stress[d] = contrib[d];
	/* :850 */
	for(int d = 0; d < 9; ++d) contrib[d] = (A[d] * mass - contrib[d] * new_dt) * d_inv;
	/* new base, re-bucket (:852-866) */
	int new_base[3], narena[3], dirv[3];
	for(int d = 0; d < 3; ++d) {
		new_base[d]	 = orc_node_index(pos[d], dx_inv) - 1;
		local_pos[d] = pos[d] - new_base[d] * dx;
		dirv[d]		 = (base_index[d] - 1) / 4 - (new_base[d] - 1) / 4;
	}
	for(int d = 0; d < 3; ++d) o->adv_cell[d] = new_base[d] - 1;
	o->dirtag = orc_dir_offset(dirv[0], dirv[1], dirv[2]);
	for(int d = 0; d < 3; ++d) {
		orc_bspline_weight(local_pos[d], dx_inv, dws[d]);
		narena[d] = (((base_index[d] - 1) & 3) + 1) + (new_base[d] - base_index[d]);
	}
	for(int d = 0; d < 9; ++d) o->contrib[d] = contrib[d];
	for(int d = 0; d < 3; ++d) {
		o->narena[d]	= narena[d];
		o->local_pos[d] = local_pos[d];
	}
	o->discarded = 0;
	if(narena[0] < 0 || narena[1] < 0 || narena[2] < 0 || narena[0] + 2 >= 8 || narena[1] + 2 >= 8 || narena[2] + 2 >= 8) {
		o->discarded = 1;
		return; /* :877-885: particle's grid contribution is discarded */
	}
	/* P2G scatter (:887-905) */
	for(int i = 0; i < 3; i++)
		for(int j = 0; j < 3; j++)
			for(int k = 0; k < 3; k++) {
				const float xp[3] = {(float) i * dx - local_pos[0], (float) j * dx - local_pos[1], (float) k * dx - local_pos[2]};
				const float W	  = dws[0][i] * dws[1][j] * dws[2][k];
				const float wm	  = mass * W;
				p2g[0][narena[0] + i][narena[1] + j][narena[2] + k] += wm;
				p2g[1][narena[0] + i][narena[1] + j][narena[2] + k] += wm * vel[0] + (contrib[0] * xp[0] + contrib[3] * xp[1] + contrib[6] * xp[2]) * W;
				p2g[2][narena[0] + i][narena[1] + j][narena[2] + k] += wm * vel[1] + (contrib[1] * xp[0] + contrib[4] * xp[1] + contrib[7] * xp[2]) * W;
				p2g[3][narena[0] + i][narena[1] + j][narena[2] + k] += wm * vel[2] + (contrib[2] * xp[0] + contrib[5] * xp[1] + contrib[8] * xp[2]) * W;
			}
}

/* ---- g2p2g, Projects/GMPM/mgmpm_kernels.cuh:665-937 ---- */
static void g2p2g_model(mpmo_ctx* c, orc_model* m, float dt, float new_dt, const int* block_list, int nlist) {
	const int r = c->rollid, n = r ^ 1;
	const orc_partition* cur  = &c->part[r];
	const orc_partition* prev = &c->part[n];
	const orc_pbuf* src		  = &m->buf[r];
	orc_pbuf* dst			  = &m->buf[n];
	const float dx = c->dx, dx_inv = c->dx_inv, d_inv = c->d_inv;
	const int nloop = block_list ? nlist : c->pbc;
	/* Serial by default (the parity tests rely on the fixed summation order).  mpmo_set_threads(n > 1) runs the particle
	 * blocks on n OpenMP threads for the multi-core cpu_baseline of bench.py: arenas become thread private, grid
	 * accumulation and cell counters atomic - the same algorithm, only the float summation order into the grid varies. */
#pragma omp parallel for schedule(dynamic, 4) num_threads(c->threads > 1 ? c->threads : 1) if(c->threads > 1)
	for(int bi = 0; bi < nloop; ++bi) {
		float g2p[3][8][8][8];
		float p2g[4][8][8][8];
		const int b		   = block_list ? block_list[bi] : bi;
		const int* blockid = cur->keys + 3 * b;
		const int size	   = dst->bucket_sizes[b];
		if(size == 0) continue;
		int nb[8];
		for(int lb = 0; lb < 8; ++lb) {
			nb[lb]			= part_query(c, cur, blockid[0] + ((lb & 4) ? 1 : 0), blockid[1] + ((lb & 2) ? 1 : 0), blockid[2] + ((lb & 1) ? 1 : 0));
			const float* gb = grid_block(c->grid[0], nb[lb]);
			for(int cx = 0; cx < 4; ++cx)
				for(int cy = 0; cy < 4; ++cy)
					for(int cz = 0; cz < 4; ++cz) {
						const int cell = cx * 16 + cy * 4 + cz;
						const int ax = cx + ((lb & 4) ? 4 : 0), ay = cy + ((lb & 2) ? 4 : 0), az = cz + ((lb & 1) ? 4 : 0);
						g2p[0][ax][ay][az] = gb[64 + cell];
						g2p[1][ax][ay][az] = gb[128 + cell];
						g2p[2][ax][ay][az] = gb[192 + cell];
					}
		}
		memset(p2g, 0, sizeof(p2g));
		for(int pidib = 0; pidib < size; ++pidib) {
			/* advection record -> source bin (:747-768) */
			const int advect = dst->blockbuckets[(size_t) b * c->ppb + pidib];
			int off[3];
			orc_dir_components(advect / c->ppb, off);
			const int source_pidib = advect & (c->ppb - 1);
			const int src_blockno  = part_query(c, prev, blockid[0] + off[0], blockid[1] + off[1], blockid[2] + off[2]);
			const float* sbin	   = bin_ptr(m, src, src->bin_offsets[src_blockno] + source_pidib / ORC_BIN);
			const int ss		   = source_pidib % ORC_BIN;
			float pos[3]		   = {sbin[ss], sbin[ORC_BIN + ss], sbin[2 * ORC_BIN + ss]};
			float J				   = (m->material == MPM_J_FLUID) ? sbin[3 * ORC_BIN + ss] : 0.f;
			/* the particle's body (:774-905): orc_particle_body below, shared with the statement-level pin of tests/golden/g16_* */
			float Fold[9] = {0};
			float log_jp  = 0.f;
			if(m->material != MPM_J_FLUID) {
				for(int d = 0; d < 9; ++d) Fold[d] = sbin[(3 + d) * ORC_BIN + ss];
				if(m->material != MPM_FIXED_COROTATED) log_jp = sbin[12 * ORC_BIN + ss];
			}
			orc_particle_out o;
			orc_particle_body(m->material, dx, dx_inv, d_inv, m->mass, m->volume, m->mu, m->lambda, m->bm, &m->p, (const float(*)[8][8][8]) g2p, p2g, pos, J, Fold, log_jp, dt, new_dt, &o);
			/* store (calculate_contribution_and_store_particle_data, :506-515, :533-548, :591-606) */
			float* dbin	 = bin_ptr(m, dst, dst->bin_offsets[b] + pidib / ORC_BIN);
			const int ds = pidib % ORC_BIN;
			dbin[ds]			   = o.pos[0];
			dbin[ORC_BIN + ds]	   = o.pos[1];
			dbin[2 * ORC_BIN + ds] = o.pos[2];
			if(m->material == MPM_J_FLUID) {
				dbin[3 * ORC_BIN + ds] = o.J;
			} else {
				for(int d = 0; d < 9; ++d) dbin[(3 + d) * ORC_BIN + ds] = o.F[d];
				if(m->nch == 13) dbin[12 * ORC_BIN + ds] = o.log_jp;
			}
			/* re-bucket (:863) */
			add_advection(c, dst, cur, o.adv_cell[0], o.adv_cell[1], o.adv_cell[2], o.dirtag, pidib);
		}
		/* arena -> next grid (:907-936) */
		for(int lb = 0; lb < 8; ++lb) {
			float* gb = grid_block(c->grid[1], nb[lb]);
			for(int ch = 0; ch < 4; ++ch)
				for(int cx = 0; cx < 4; ++cx)
					for(int cy = 0; cy < 4; ++cy)
						for(int cz = 0; cz < 4; ++cz) {
							const float add = p2g[ch][cx + ((lb & 4) ? 4 : 0)][cy + ((lb & 2) ? 4 : 0)][cz + ((lb & 1) ? 4 : 0)];
#pragma omp atomic
							gb[ch * 64 + cx * 16 + cy * 4 + cz] += add;
						}
		}
	}
}

int mpmo_g2p2g(mpmo_ctx* c, float dt, float next_dt) {
	if(!c || !c->ready) return MPM_ERR_NOT_READY;
	double t0	= now_ms();
	const int n = c->rollid ^ 1;
	clear_grid(c, c->grid[1], c->nbc); /* gmpm_simulator.cuh:383 */
	for(int mi = 0; mi < c->nmodels; ++mi) {
		orc_model* m = &c->models[mi];
		par_fill(c, m->buf[n].cell_counts, 0, sizeof(int) * (size_t) c->ebc * ORC_BLOCKVOL); /* :389 */
		if((size_t) m->bincount > m->buf[n].bin_cap) return fail(c, MPM_ERR_CAPACITY, "bin capacity");
		g2p2g_model(c, m, dt, next_dt, NULL, 0);
	}
	c->timers.g2p2g_ms = (float) (now_ms() - t0);
	return MPM_OK;
}

/* ---- MGSP halo path: Projects/MGSP/halo_kernels.cuh, mgsp_benchmark.cuh:421-465, :661-776 ---- */
int mpmo_halo_keys(mpmo_ctx* c, int* keys, int capacity_blocks, int* count) {
	if(!c || !c->ready) return MPM_ERR_NOT_READY;
	const int ncopy = c->nbc < capacity_blocks ? c->nbc : capacity_blocks;
	memcpy(keys, c->part[c->rollid].keys, sizeof(int) * 3 * (size_t) ncopy);
	if(count) *count = c->nbc;
	return MPM_OK;
}
int mpmo_halo_tag_begin(mpmo_ctx* c) {
	if(!c || !c->ready) return MPM_ERR_NOT_READY;
	if(!c->overlap) {
		c->overlap	  = (int*) calloc(c->cap + 1, sizeof(int));
		c->halo_list  = (int*) calloc(c->cap + 1, sizeof(int));
		c->inner_list = (int*) calloc(c->cap + 1, sizeof(int));
	}
	memset(c->overlap, 0, sizeof(int) * ((size_t) c->nbc + 1));
	for(int p = 0; p < 32; ++p) c->send_count[p] = 0;
	c->halo_tagged = 0;
	return MPM_OK;
}
/* mark_overlapping_blocks, halo_kernels.cuh:21-35 (restricted to blocks that are neighbor blocks on both sides) */
int mpmo_halo_tag_peer(mpmo_ctx* c, int peer, const int* peer_keys, int n) {
	if(!c || !c->ready || !c->overlap || peer < 0 || peer >= 32) return MPM_ERR_INVALID;
	if(!c->send_ids[peer]) c->send_ids[peer] = (int*) calloc(c->cap + 1, sizeof(int));
	for(int i = 0; i < n; ++i) {
		const int b = part_query(c, &c->part[c->rollid], peer_keys[3 * i], peer_keys[3 * i + 1], peer_keys[3 * i + 2]);
		if(b >= 0 && b < c->nbc) {
			c->overlap[b] |= 1 << peer;
			c->send_ids[peer][c->send_count[peer]++] = b;
		}
	}
	return MPM_OK;
}
/* collect_blockids_for_halo_reduction, halo_kernels.cuh:37-62 */
int mpmo_halo_tag_end(mpmo_ctx* c, int* halo_particle_blocks, int* send_counts) {
	if(!c || !c->ready || !c->overlap) return MPM_ERR_INVALID;
	const orc_partition* P = &c->part[c->rollid];
	c->n_halo = c->n_inner = 0;
	for(int b = 0; b < c->pbc; ++b) {
		int halo = 0;
		for(int i = 0; i < 2; ++i)
			for(int j = 0; j < 2; ++j)
				for(int k = 0; k < 2; ++k) {
					const int nb = part_query(c, P, P->keys[3 * b] + i, P->keys[3 * b + 1] + j, P->keys[3 * b + 2] + k);
					if(nb >= 0 && c->overlap[nb]) halo = 1;
				}
		if(halo)
			c->halo_list[c->n_halo++] = b;
		else
			c->inner_list[c->n_inner++] = b;
	}
	if(halo_particle_blocks) *halo_particle_blocks = c->n_halo;
	if(send_counts)
		for(int p = 0; p < 32; ++p) send_counts[p] = c->send_count[p];
	c->halo_tagged = 1;
	return MPM_OK;
}
int mpmo_g2p2g_halo(mpmo_ctx* c, float dt, float next_dt) {
	if(!c || !c->ready || !c->halo_tagged) return MPM_ERR_NOT_READY;
	const int n = c->rollid ^ 1;
	clear_grid(c, c->grid[1], c->nbc);
	for(int mi = 0; mi < c->nmodels; ++mi) {
		orc_model* m = &c->models[mi];
		memset(m->buf[n].cell_counts, 0, sizeof(int) * (size_t) c->ebc * ORC_BLOCKVOL);
		g2p2g_model(c, m, dt, next_dt, c->halo_list, c->n_halo);
	}
	return MPM_OK;
}
int mpmo_g2p2g_interior(mpmo_ctx* c, float dt, float next_dt) {
	if(!c || !c->ready || !c->halo_tagged) return MPM_ERR_NOT_READY;
	for(int mi = 0; mi < c->nmodels; ++mi) g2p2g_model(c, &c->models[mi], dt, next_dt, c->inner_list, c->n_inner);
	return MPM_OK;
}
/* collect_grid_blocks, halo_kernels.cuh:64-80 */
int mpmo_halo_collect(mpmo_ctx* c, int peer, int gid, int* keys, float* blocks, int capacity_blocks, int* nsend) {
	if(!c || !c->ready || !c->halo_tagged || peer < 0 || peer >= 32 || gid < 0 || gid > 1) return MPM_ERR_INVALID;
	const int n = c->send_count[peer];
	if(nsend) *nsend = n;
	if(n > capacity_blocks) return fail(c, MPM_ERR_CAPACITY, "halo send buffer too small");
	for(int i = 0; i < n; ++i) {
		const int b = c->send_ids[peer][i];
		memcpy(keys + 3 * i, c->part[c->rollid].keys + 3 * b, sizeof(int) * 3);
		memcpy(blocks + (size_t) i * 256, grid_block(c->grid[gid], b), sizeof(float) * 256);
	}
	return MPM_OK;
}
/* reduce_grid_blocks, halo_kernels.cuh:82-97 */
int mpmo_halo_reduce(mpmo_ctx* c, int gid, const int* keys, const float* blocks, int nrecv) {
	if(!c || !c->ready || gid < 0 || gid > 1) return MPM_ERR_INVALID;
	for(int i = 0; i < nrecv; ++i) {
		const int b = part_query(c, &c->part[c->rollid], keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]);
		if(b < 0 || b >= c->nbc) continue;
		float* g = grid_block(c->grid[gid], b);
		for(int k = 0; k < 256; ++k) g[k] += blocks[(size_t) i * 256 + k];
	}
	return MPM_OK;
}

/* ---- partition rebuild, Projects/GMPM/gmpm_simulator.cuh:415-579 ---- */
int mpmo_rebuild_partition(mpmo_ctx* c, mpm_counts* counts) {
	if(!c || !c->ready) return MPM_ERR_NOT_READY;
	double t0	= now_ms();
	const int r = c->rollid, n = r ^ 1;
	const int ebc = c->ebc, nbc = c->nbc;
	/* cell_bucket_to_block over exterior blocks (:421-432) */
	for(int mi = 0; mi < c->nmodels; ++mi) {
		orc_pbuf* b = &c->models[mi].buf[n];
		memset(b->bucket_sizes, 0, sizeof(int) * ((size_t) ebc + 1));
		cell_bucket_to_block(c, b, ebc);
		b->bucket_sizes[ebc] = 0;
	}
	/* mark_active_grid_blocks (mgmpm_kernels.cuh:939-952) */
	memset(c->marks, 0, sizeof(int) * (size_t) nbc);
	ORC_PAR
	for(int b = 0; b < nbc; ++b) {
		const float* g = grid_block(c->grid[1], b);
		for(int cell = 0; cell < 64; ++cell)
			if(g[cell] != 0.0f) {
				c->marks[b] = 1;
				break;
			}
	}
	/* mark_active_particle_blocks (:954-964) OR-ed over models, exclusive scan, inverse map */
	memset(c->sources, 0, sizeof(int) * ((size_t) ebc + 1));
	for(int mi = 0; mi < c->nmodels; ++mi) {
		const orc_pbuf* b = &c->models[mi].buf[n];
		for(int blk = 0; blk <= ebc; ++blk)
			if(b->bucket_sizes[blk] > 0) c->sources[blk] = 1;
	}
	{
		int acc = 0;
		for(int blk = 0; blk <= ebc; ++blk) {
			c->destinations[blk] = acc;
			acc += c->sources[blk];
		}
	}
	const int new_pbc = c->destinations[ebc];
	/* exclusive_scan_inverse, Library/MnBase/Algorithm/MappingKernels.cuh:44-55 */
	for(int blk = 0; blk < ebc; ++blk)
		if(c->destinations[blk] != c->destinations[blk + 1]) c->sources[c->destinations[blk]] = blk;
	if((size_t) new_pbc > c->cap) return fail(c, MPM_ERR_CAPACITY, "Too much active blocks");
	/* update_partition (mgmpm_kernels.cuh:966-977) */
	orc_partition* Pn		= &c->part[n];
	const orc_partition* Pr = &c->part[r];
	part_reset_table(c, Pn);
	Pn->count = new_pbc;
	for(int b = 0; b < new_pbc; ++b) {
		const int s			= c->sources[b];
		Pn->keys[3 * b]		= Pr->keys[3 * s];
		Pn->keys[3 * b + 1] = Pr->keys[3 * s + 1];
		Pn->keys[3 * b + 2] = Pr->keys[3 * s + 2];
		Pn->index_table[key_index(c, Pn->keys[3 * b], Pn->keys[3 * b + 1], Pn->keys[3 * b + 2])] = b;
	}
	/* update_buckets (:979-1000) + bin offsets (:493-505) */
	for(int mi = 0; mi < c->nmodels; ++mi) {
		orc_model* m	   = &c->models[mi];
		const orc_pbuf* bn = &m->buf[n];
		orc_pbuf* br	   = &m->buf[r];
		ORC_PAR
		for(int b = 0; b < new_pbc; ++b) {
			const int s			= c->sources[b];
			const int cnt		= bn->bucket_sizes[s];
			br->bucket_sizes[b] = cnt;
			memcpy(br->blockbuckets + (size_t) b * c->ppb, bn->blockbuckets + (size_t) s * c->ppb, sizeof(int) * (size_t) cnt);
		}
		br->bucket_sizes[new_pbc] = 0;
		m->bincount				  = bin_offsets_scan(c, br, new_pbc);
		if((size_t) m->bincount > br->bin_cap) return fail(c, MPM_ERR_CAPACITY, "bin capacity");
	}
	/* neighbors (:513-526) */
	register_neighbor_blocks(c, Pn, new_pbc);
	const int new_nbc = Pn->count;
	if((size_t) new_nbc > c->cap) return fail(c, MPM_ERR_CAPACITY, "Too much neighbour blocks");
	/* clear grid[0], copy_selected_grid_blocks (:536-541, kernel mgmpm_kernels.cuh:1002-1020) */
	clear_grid(c, c->grid[0], ebc > new_nbc ? ebc : new_nbc);
	ORC_PAR
	for(int b = 0; b < nbc; ++b) {
		if(!c->marks[b]) continue;
		const int bno = part_query(c, Pn, Pr->keys[3 * b], Pr->keys[3 * b + 1], Pr->keys[3 * b + 2]);
		if(bno == -1) continue;
		memcpy(grid_block(c->grid[0], bno), grid_block(c->grid[1], b), sizeof(float) * 256);
	}
	/* exterior (:560-570) */
	register_exterior_blocks(c, Pn, new_pbc);
	const int new_ebc = Pn->count;
	if((size_t) new_ebc > c->cap) return fail(c, MPM_ERR_CAPACITY, "Too much exterior blocks");
	c->pbc	  = new_pbc;
	c->nbc	  = new_nbc;
	c->ebc	  = new_ebc;
	c->rollid = n; /* :578 */
	c->timers.partition_ms = (float) (now_ms() - t0);
	if(counts) {
		memset(counts, 0, sizeof(*counts));
		counts->particle_blocks = c->pbc;
		counts->neighbor_blocks = c->nbc;
		counts->exterior_blocks = c->ebc;
		counts->model_count		= c->nmodels;
		for(int mi = 0; mi < c->nmodels; ++mi) {
			counts->bins[mi] = c->models[mi].bincount;
			int64_t np		 = 0;
			for(int b = 0; b < c->pbc; ++b) np += c->models[mi].buf[c->rollid ^ 1].bucket_sizes[b];
			counts->particles[mi] = np;
		}
	}
	return MPM_OK;
}

int mpmo_get_counts(mpmo_ctx* c, mpm_counts* counts) {
	if(!c || !c->ready || !counts) return MPM_ERR_NOT_READY;
	memset(counts, 0, sizeof(*counts));
	counts->particle_blocks = c->pbc;
	counts->neighbor_blocks = c->nbc;
	counts->exterior_blocks = c->ebc;
	counts->model_count		= c->nmodels;
	for(int mi = 0; mi < c->nmodels; ++mi) {
		counts->bins[mi] = c->models[mi].bincount;
		int64_t np		 = 0;
		for(int b = 0; b < c->pbc; ++b) np += c->models[mi].buf[c->rollid ^ 1].bucket_sizes[b];
		counts->particles[mi] = np;
	}
	return MPM_OK;
}

int mpmo_substep(mpmo_ctx* c, float dt, float step_time, float frame_time, float dt_default, float* next_dt, float* max_vel) {
	float mv2 = 0.f;
	int rc	  = mpmo_grid_update(c, dt, &mv2);
	if(rc) return rc;
	if(isinf(mv2)) return fail(c, MPM_ERR_NONFINITE, "Maximum velocity is infinity");
	const float mv = sqrtf(mv2); /* gmpm_simulator.cuh:360 */
	const float nd = mpmo_compute_dt(c, mv, step_time, frame_time, dt_default);
	if(max_vel) *max_vel = mv;
	if(next_dt) *next_dt = nd;
	rc = mpmo_g2p2g(c, dt, nd);
	if(rc) return rc;
	rc = mpmo_rebuild_partition(c, NULL);
	c->timers.total_ms = c->timers.grid_update_ms + c->timers.g2p2g_ms + c->timers.partition_ms;
	return rc;
}

int mpmo_run_fixed(mpmo_ctx* c, int nsteps, float dt) {
	for(int s = 0; s < nsteps; ++s) {
		float mv2 = 0.f;
		int rc	  = mpmo_grid_update(c, dt, &mv2);
		if(rc) return rc;
		if(isinf(mv2)) return fail(c, MPM_ERR_NONFINITE, "Maximum velocity is infinity");
		rc = mpmo_g2p2g(c, dt, dt);
		if(rc) return rc;
		rc = mpmo_rebuild_partition(c, NULL);
		if(rc) return rc;
	}
	return MPM_OK;
}

/* the oracle carries the reference's F (particle_buffer.cuh:141-264); the HIP library answers MPM_STATE_B */
int mpmo_state_kind(void) {
	return 0;
}

/* retrieve_particle_buffer, mgmpm_kernels.cuh:1087-1122 (+ state for the parity tests) */
int mpmo_retrieve_state(mpmo_ctx* c, int model, float* xyz, float* state9, float* logjp, size_t* n) {
	if(!c || !c->ready || model < 0 || model >= c->nmodels || !n) return MPM_ERR_INVALID;
	const int r = c->rollid, nn = r ^ 1;
	orc_model* m		= &c->models[model];
	const orc_pbuf* cur = &m->buf[r];
	const orc_pbuf* nxt = &m->buf[nn];
	size_t count		= 0;
	for(int b = 0; b < c->pbc; ++b) {
		const int* key = c->part[r].keys + 3 * b;
		for(int pidib = 0; pidib < nxt->bucket_sizes[b]; ++pidib) {
			const int advect = nxt->blockbuckets[(size_t) b * c->ppb + pidib];
			int off[3];
			orc_dir_components(advect / c->ppb, off);
			const int sp   = advect % c->ppb;
			const int sblk = part_query(c, &c->part[nn], key[0] + off[0], key[1] + off[1], key[2] + off[2]);
			const float* bin = bin_ptr(m, cur, cur->bin_offsets[sblk] + sp / ORC_BIN);
			const int s		 = sp % ORC_BIN;
			if(count >= *n) return fail(c, MPM_ERR_CAPACITY, "output array too small");
			if(xyz) {
				xyz[3 * count]	   = bin[s];
				xyz[3 * count + 1] = bin[ORC_BIN + s];
				xyz[3 * count + 2] = bin[2 * ORC_BIN + s];
			}
			if(state9) {
				if(m->material == MPM_J_FLUID) {
					state9[9 * count] = bin[3 * ORC_BIN + s];
					for(int d = 1; d < 9; ++d) state9[9 * count + d] = 0.f;
				} else {
					for(int d = 0; d < 9; ++d) state9[9 * count + d] = bin[(3 + d) * ORC_BIN + s];
				}
			}
			if(logjp) logjp[count] = (m->nch == 13) ? bin[12 * ORC_BIN + s] : 0.f;
			count++;
		}
	}
	*n = count;
	return MPM_OK;
}

int mpmo_retrieve_positions(mpmo_ctx* c, int model, float* xyz, size_t* n) {
	return mpmo_retrieve_state(c, model, xyz, NULL, NULL, n);
}

int mpmo_get_timers(mpmo_ctx* c, mpm_timers* t) {
	if(!c || !t) return MPM_ERR_INVALID;
	*t = c->timers;
	return MPM_OK;
}

int mpmo_grid_totals(mpmo_ctx* c, double out[4]) {
	if(!c || !c->ready) return MPM_ERR_NOT_READY;
	for(int ch = 0; ch < 4; ++ch) out[ch] = 0.0;
	for(int b = 0; b < c->nbc; ++b) {
		const float* g = grid_block(c->grid[0], b);
		for(int ch = 0; ch < 4; ++ch)
			for(int cell = 0; cell < 64; ++cell) out[ch] += g[ch * 64 + cell];
	}
	return MPM_OK;
}

int mpmo_dump_grid(mpmo_ctx* c, int* keys, float* blocks, size_t* nblocks) {
	if(!c || !c->ready || !nblocks) return MPM_ERR_NOT_READY;
	if(*nblocks < (size_t) c->nbc) return fail(c, MPM_ERR_CAPACITY, "grid dump too small");
	if(keys) memcpy(keys, c->part[c->rollid].keys, sizeof(int) * 3 * (size_t) c->nbc);
	if(blocks) memcpy(blocks, c->grid[0], sizeof(float) * 256 * (size_t) c->nbc);
	*nblocks = c->nbc;
	return MPM_OK;
}

/* fused substep API (include/claymore_amd.h): synchronous here, so built from the phase functions */
int mpmo_mgsp_begin(mpmo_ctx* c, float dt, float next_dt) {
	int rc = mpmo_grid_update(c, dt, &c->last_max_vel_sqr);
	if(rc) return rc;
	return mpmo_g2p2g_halo(c, dt, next_dt);
}
int mpmo_mgsp_rebuild_export(mpmo_ctx* c, int* keys, int pad_rows) {
	int rc = mpmo_rebuild_partition(c, NULL);
	if(rc) return rc;
	memset(keys, 0, sizeof(int) * 3 * (size_t) pad_rows);
	keys[0]		= c->nbc;
	const int n = c->nbc < pad_rows - 1 ? c->nbc : pad_rows - 1;
	memcpy(keys + 3, c->part[c->rollid].keys, sizeof(int) * 3 * (size_t) n);
	return MPM_OK;
}
int mpmo_mgsp_tag(mpmo_ctx* c, const int* all_keys, int pad_rows, int world, int rank) {
	int rc = mpmo_halo_tag_begin(c);
	if(rc) return rc;
	c->peer_rows_max = 0;
	for(int p = 0; p < world; ++p) {
		const int* rows = all_keys + (size_t) 3 * pad_rows * p;
		if(rows[0] + 1 > c->peer_rows_max) c->peer_rows_max = rows[0] + 1;
		if(p == rank) continue;
		const int n = rows[0] < pad_rows - 1 ? rows[0] : pad_rows - 1;
		rc			= mpmo_halo_tag_peer(c, p, rows + 3, n);
		if(rc) return rc;
	}
	return mpmo_halo_tag_end(c, NULL, NULL);
}
int mpmo_mgsp_end(mpmo_ctx* c, int* send_counts, int* halo_particle_blocks, int* max_peer_rows, float* max_vel_sqr) {
	if(!c || !c->ready) return MPM_ERR_NOT_READY;
	if(send_counts)
		for(int p = 0; p < 32; ++p) send_counts[p] = c->send_count[p];
	if(halo_particle_blocks) *halo_particle_blocks = c->n_halo;
	if(max_peer_rows) *max_peer_rows = c->peer_rows_max;
	if(max_vel_sqr) *max_vel_sqr = c->last_max_vel_sqr;
	return MPM_OK;
}

/* table consistency: the reference's check_table debug kernel, mgmpm_kernels.cuh:1022-1032 */
int mpmo_check_table(mpmo_ctx* c) {
	if(!c || !c->ready) return -1;
	const orc_partition* P = &c->part[c->rollid];
	int bad				   = 0;
	for(int b = 0; b < c->ebc; ++b)
		if(part_query(c, P, P->keys[3 * b], P->keys[3 * b + 1], P->keys[3 * b + 2]) != b) bad++;
	return bad;
}

/* ---- G20 hooks (tests/test_oracle_golden.py): the integer bookkeeping of the particle path against the reference's own statements ----
 * book_dump: the partition (keys in block order, {particle, neighbour, exterior} block counts) and one buffer's buckets (sizes, entries, bin
 * offsets); which = 0: buf[rollid] (what initial_setup filled: entries are particle ids), 1: buf[rollid ^ 1] (what the last rebuild filled:
 * entries are (dirtag * ppb) | particle_id_in_block of the source block).
 * book_advect: what g2p2g hands to add_advection (mgmpm_kernels.cuh:852-866) without the physics - every bucketed particle's cell moved by
 * delta[particle id] in {-1,0,1}^3; valid right after initial_setup (the buckets of buf[rollid] still hold particle ids). */
int mpmo_fn_book_dump(mpmo_ctx* c, int model, int which, int* counts3, int* keys, int* sizes, int* buckets, int* binoff) {
	if(!c || !c->ready || model < 0 || model >= c->nmodels) return MPM_ERR_NOT_READY;
	const orc_partition* P = &c->part[c->rollid];
	const orc_pbuf* b	   = &c->models[model].buf[which ? c->rollid ^ 1 : c->rollid];
	counts3[0] = c->pbc, counts3[1] = c->nbc, counts3[2] = c->ebc;
	memcpy(keys, P->keys, sizeof(int) * 3 * (size_t) c->ebc);
	size_t o = 0;
	for(int blk = 0; blk < c->pbc; ++blk) {
		sizes[blk] = b->bucket_sizes[blk];
		for(int i = 0; i < b->bucket_sizes[blk]; ++i) buckets[o++] = b->blockbuckets[(size_t) blk * c->ppb + i];
	}
	for(int blk = 0; blk <= c->pbc; ++blk) binoff[blk] = b->bin_offsets[blk];
	return MPM_OK;
}
int mpmo_fn_book_advect(mpmo_ctx* c, int model, const int* delta) {
	if(!c || !c->ready || model < 0 || model >= c->nmodels) return MPM_ERR_NOT_READY;
	const int r = c->rollid, n = r ^ 1;
	orc_model* m	   = &c->models[model];
	const orc_pbuf* br = &m->buf[r];
	orc_pbuf* bn	   = &m->buf[n];
	memset(bn->cell_counts, 0, sizeof(int) * (size_t) c->ebc * ORC_BLOCKVOL);
	for(int blk = 0; blk < c->pbc; ++blk)
		for(int pidib = 0; pidib < br->bucket_sizes[blk]; ++pidib) {
			const int pid = br->blockbuckets[(size_t) blk * c->ppb + pidib];
			int cell[3], nc[3], bd[3];
			for(int d = 0; d < 3; ++d) {
				cell[d] = orc_node_index(m->xyz[3 * (size_t) pid + d], c->dx_inv) - 2; /* base_index - 1, :774-777 */
				nc[d]	= cell[d] + delta[3 * (size_t) pid + d];
				bd[d]	= cell[d] / 4 - nc[d] / 4; /* :860-862 (C++ truncating division) */
			}
			add_advection(c, bn, &c->part[r], nc[0], nc[1], nc[2], orc_dir_offset(bd[0], bd[1], bd[2]), pidib);
		}
	return MPM_OK;
}

/* ---- function-level entry points for the golden-vector tests ---- */
void mpmo_fn_bspline(const float* p, size_t n, float dx_inv, float* out3) {
	for(size_t i = 0; i < n; ++i) orc_bspline_weight(p[i], dx_inv, out3 + 3 * i);
}
void mpmo_fn_node_index(const float* x, size_t n, float dx_inv, int* out) {
	for(size_t i = 0; i < n; ++i) out[i] = orc_node_index(x[i], dx_inv);
}
int mpmo_fn_dir_offset(int dx, int dy, int dz) {
	return orc_dir_offset(dx, dy, dz);
}
void mpmo_fn_dir_components(int dir, int d[3]) {
	orc_dir_components(dir, d);
}
float mpmo_fn_compute_dt(float max_vel, float cur, float next, float dt_default, float dx, float cfl) {
	return orc_compute_dt(max_vel, cur, next, dt_default, dx, cfl);
}
float mpmo_fn_compute_dt_mgsp(float max_vel, float cur, float next, float dt_default, float dx) {
	return orc_compute_dt_mgsp(max_vel, cur, next, dt_default, dx);
}
/* the collision object's functions one by one (golden vectors G13-G15, Projects/MGSP/boundary_condition.cuh:67-248) */
void mpmo_fn_rot_angle_to_matrix(float omega, int dim, float* out9) {
	col_rot_angle_to_matrix(omega, dim, out9);
}
/* x[n*3] -> out6 = {inside the wall-free zone, query_sdf, sdis, normal[3]} (sdis / normal only where inside) */
int mpmo_fn_query_sdf(const mpmo_ctx* c, const float* x, size_t n, float* out6) {
	if(!c || !c->has_collision) return MPM_ERR_NOT_READY;
	const float lo = (float) c->cfg.boundary_blocks * c->dx * 4.f;
	const float hi = (float) (c->G - c->cfg.boundary_blocks) * 4.f * c->dx;
	for(size_t i = 0; i < n; ++i) {
		const float* p = x + 3 * i;
		float* o	   = out6 + 6 * i;
		float nq[3]	   = {0.f, 0.f, 0.f};
		const int inside = !(p[0] < lo || p[0] >= hi || p[1] < lo || p[1] >= hi || p[2] < lo || p[2] >= hi);
		o[0] = (float) inside;
		o[1] = (float) col_query_sdf(c, nq, p);
		o[2] = o[3] = o[4] = o[5] = 0.f;
		if(inside) o[2] = col_signed_distance_and_normal(c, p, o + 3);
	}
	return MPM_OK;
}
/* nodes[n*3] (global node indices), vel[n*3] in place: detect_and_resolve_collision at `time` with the installed object */
int mpmo_fn_collision_resolve(const mpmo_ctx* c, const int* nodes, size_t n, float time, float* vel) {
	if(!c || !c->has_collision) return MPM_ERR_NOT_READY;
	for(size_t i = 0; i < n; ++i) {
		const int block_id[3] = {nodes[3 * i] / 4, nodes[3 * i + 1] / 4, nodes[3 * i + 2] / 4};
		const int cell_id[3]  = {nodes[3 * i] - block_id[0] * 4, nodes[3 * i + 1] - block_id[1] * 4, nodes[3 * i + 2] - block_id[2] * 4};
		col_detect_and_resolve(c, block_id, cell_id, time, vel + 3 * i);
	}
	return MPM_OK;
}
void mpmo_fn_mat(const float* a, const float* b, const float* diag, float* out36) {
	float e[9];
	orc_matmul3(a, b, out36);
	orc_mat_diag_matT(out36 + 9, a, diag, b);
	orc_mat_matT(a, e);
	memcpy(out36 + 18, e, sizeof(e));
	orc_deviatoric(e, out36 + 27);
}
/* math::svd itself (oracle only: golden vector G3) */
int mpmo_test_svd(const float* F, size_t n, float* out21, int device) {
	(void) device;
	for(size_t i = 0; i < n; ++i) orc_svd3(F + 9 * i, out21 + 21 * i, out21 + 21 * i + 9, out21 + 21 * i + 12);
	return MPM_OK;
}
/* the C ABI's view of the same decomposition: F F^T = U diag(S^2) U^T */
int mpmo_test_eig(const float* F, size_t n, float* out12, int device) {
	(void) device;
	for(size_t i = 0; i < n; ++i) {
		float U[9], S[3], V[9];
		orc_svd3(F + 9 * i, U, S, V);
		memcpy(out12 + 12 * i, U, sizeof(U));
		for(int k = 0; k < 3; ++k) out12[12 * i + 9 + k] = S[k] * S[k];
	}
	return MPM_OK;
}
int mpmo_test_stress(int material, const mpm_material_params* p, const float* Fin, const float* logjp, size_t n, float* out19, int device) {
	(void) device;
	const float e = p->youngs_modulus, nu = p->poisson_ratio;
	const float lambda = e * nu / ((1 + nu) * (1 - 2 * nu));
	const float mu	   = e / (2 * (1 + nu));
	const float bm	   = 2.f / 3.f * (e / (2 * (1 + nu))) + (e * nu / ((1 + nu) * (1 - 2 * nu)));
	for(size_t i = 0; i < n; ++i) {
		float F[9], PF[9];
		memcpy(F, Fin + 9 * i, sizeof(F));
		float lj = logjp ? logjp[i] : 0.f;
		switch(material) {
			case MPM_FIXED_COROTATED: orc_stress_fixed_corotated(p->volume, mu, lambda, F, PF); break;
			case MPM_SAND: orc_stress_sand(p->volume, mu, lambda, p->cohesion, p->beta, p->yield_surface, p->volume_correction, F, &lj, PF); break;
			case MPM_NACC: orc_stress_nacc(p->volume, mu, lambda, bm, p->xi, p->beta, p->msqr, p->hardening_on, F, &lj, PF); break;
			default: return MPM_ERR_INVALID;
		}
		memcpy(out19 + 19 * i, F, sizeof(F));
		memcpy(out19 + 19 * i + 9, PF, sizeof(PF));
		out19[19 * i + 18] = lj;
	}
	return MPM_OK;
}
/* G19: rows (mass, mvx, mvy, mvz, is_in_bound bits x=4 y=2 z=1) -> (vx, vy, vz as the cell holds them afterwards, vel_sqr) */
void mpmo_fn_grid_cells(const float* in5, size_t n, float gravity, float dt, float* out4) {
	for(size_t i = 0; i < n; ++i) {
		float g[256] = {0};
		const int key[3] = {0, 0, 0};
		for(int ch = 0; ch < 4; ++ch) g[64 * ch] = in5[5 * i + ch];
		const int bound = (int) in5[5 * i + 4];
		out4[4 * i + 3] = orc_grid_cell(NULL, key, 0, g, bound & 4, bound & 2, bound & 1, gravity, dt);
		for(int d = 0; d < 3; ++d) out4[4 * i + d] = g[64 * (1 + d)];
	}
}
/* G16-G18: one particle per row through the body of g2p2g on a given velocity arena (g2pbuffer[3][8][8][8]) and an empty scatter arena.
 * rows_in: pos[3], F[9] (column-major; J-fluid: J in F[0]), log_jp (13 floats); out_f (154 floats): vel[3] A[9] pos[3] | F[9] log_jp stress[9] | contrib[9]
 * local_pos[3] | the 27 x {m, mvx, mvy, mvz} the particle left in the arena, stencil offsets row-major; out_i (14 ints): base[3] arena[3]
 * add_advection cell[3] dirtag narena[3] discarded (narena = -99 when discarded) */
int mpmo_fn_particle_step(int material, const mpm_material_params* p, int domain_bits, const float* arena, const float* rows_in, size_t n, float dt, float new_dt, float* out_f, int* out_i) {
	if(!p || !arena || !rows_in || !out_f || !out_i || material < 0 || material > 3) return MPM_ERR_INVALID;
	const float dx_inv = (float) (1 << domain_bits), dx = 1.f / dx_inv, d_inv = 4.f * dx_inv * dx_inv; /* settings.h:60-66 */
	const float e = p->youngs_modulus, nu = p->poisson_ratio;
	const float mass = p->volume * p->rho, lambda = e * nu / ((1 + nu) * (1 - 2 * nu)), mu = e / (2 * (1 + nu));
	const float bm = 2.f / 3.f * (e / (2 * (1 + nu))) + (e * nu / ((1 + nu) * (1 - 2 * nu)));
	for(size_t i = 0; i < n; ++i) {
		float p2g[4][8][8][8];
		memset(p2g, 0, sizeof(p2g));
		const float* in = rows_in + 13 * i;
		orc_particle_out o;
		orc_particle_body(material, dx, dx_inv, d_inv, mass, p->volume, mu, lambda, bm, p, (const float(*)[8][8][8]) arena, p2g, in, in[3], in + 3, in[12], dt, new_dt, &o); /* (J-fluid: the state J sits in the slot of F[0]) */
		if(material == MPM_J_FLUID) o.F[0] = o.J;
		float* f = out_f + 154 * i;
		int* q	 = out_i + 14 * i;
		memcpy(f, o.vel, 12), memcpy(f + 3, o.A, 36), memcpy(f + 12, o.pos, 12);
		memcpy(f + 15, o.F, 36), f[24] = o.log_jp, memcpy(f + 25, o.stress, 36);
		memcpy(f + 34, o.contrib, 36), memcpy(f + 43, o.local_pos, 12);
		for(int s = 0; s < 27; ++s)
			for(int ch = 0; ch < 4; ++ch) f[46 + 4 * s + ch] = o.discarded ? 0.f : p2g[ch][o.narena[0] + s / 9][o.narena[1] + (s / 3) % 3][o.narena[2] + s % 3];
		memcpy(q, o.base, 12), memcpy(q + 3, o.arena, 12), memcpy(q + 6, o.adv_cell, 12);
		q[9] = o.dirtag;
		for(int d = 0; d < 3; ++d) q[10 + d] = o.discarded ? -99 : o.narena[d];
		q[13] = o.discarded;
	}
	return MPM_OK;
}
/* J-fluid inline block: in = J, A[9]; out10 = J', contrib[9] */
void mpmo_fn_jfluid(const float* J, const float* A, size_t n, float dt, float d_inv, float volume, float bulk, float gamma, float viscosity, float* out10) {
	for(size_t i = 0; i < n; ++i) out10[10 * i] = orc_jfluid(J[i], A + 9 * i, dt, d_inv, volume, bulk, gamma, viscosity, out10 + 10 * i + 1);
}
