"use strict"
// Row tiling over the GPUs of one node, one Node process per GPU (SURVEY.md §8e; Python twin: rfx_amd/tiling.py).
// TiledRenderer wraps the tile's Renderer and performs the exchange steps through the C ABI's RCCL entry points
// (include/rfx.h "row-tiled runs": rfx_halo_exchange / rfx_allgather_history / rfx_comm_wait):
//   after K2 and after every K3 pass : halo Send/Recv of the textures just written with the row neighbours, issued on the
//                                      context's exchange stream; the next draw (K3 pass, K4) first produces the tile INTERIOR
//                                      (setRowWindow), then waits, then draws the two boundary strips
//   the composed GI                  : next frame's K1 gathers it anywhere on screen, but only its SHADING half reads it, so K1 runs as
//                                      ssgiTrace / ssgiShade: historyGather "all" (default) all-gathers .rgb (RFX_TEX_COMPOSE_RGB)
//                                      after K4 and waits between the two halves; "bounded" moves, between the two, exactly the rows
//                                      the traced rays will read, from their owners (rfx_gather_history_rows); "peer" (usePeerHistory)
//                                      has this rank's own kernel load those column blocks out of the owners' planes through HIP IPC
//                                      mappings (rfx_peer_*): no collective, no host wait
const addon = require("../napi/rfx_napi.node")
const { Renderer, TEX } = require("./Renderer")

function splitRows(height, nranks, rank) {
	return addon.splitRows(height, nranks, rank)
}

// rows of halo that make the tiled result exact (rfx_amd/tiling.py required_halo)
function requiredHalo(radius, maxAbsVelocityY, frameHeight, frameWidth) {
	const aspectRows = frameWidth ? Math.max(1, frameHeight / frameWidth) : 1
	const k3 = Math.ceil(radius * aspectRows) + 2
	const k2 = Math.ceil(Math.abs(maxAbsVelocityY) * frameHeight) + 4
	return Math.max(k3, k2, 2)
}

class TiledRenderer {
	// `uniqueId`: the 128-byte Buffer of commUniqueId() made by rank 0 and handed to every process
	constructor(width, height, rank, nranks, haloRows, uniqueId, options) {
		options = options || {}
		const t = splitRows(height, nranks, rank)
		this.rank = rank
		this.nranks = nranks
		// options.inner / options.comm: a stand-in tile renderer and exchange layer (tests record the call sequence without a GPU)
		this.inner = options.inner || new Renderer(width, height, Object.assign({}, options, { tileY0: t[0], tileRows: t[1], haloRows: nranks > 1 ? haloRows : 0 }))
		this._comm = options.comm || {
			haloExchange: (tex, up, down) => addon.haloExchange(this.inner._h, tex, up, down),
			allgatherHistory: tex => addon.allgatherHistory(this.inner._h, tex),
			gatherHistoryRows: tex => addon.gatherHistoryRows(this.inner._h, tex),
			peerExport: tex => addon.peerExport(this.inner._h, tex),
			peerOpen: (tex, blobs, rank, nranks) => addon.peerOpen(this.inner._h, tex, blobs, rank, nranks),
			peerGatherHistory: tex => addon.peerGatherHistory(this.inner._h, tex),
			peerClose: () => addon.peerClose(this.inner._h),
			commWait: () => addon.commWait(this.inner._h)
		}
		this.width = width
		this.height = height
		this.tileY0 = t[0]
		this.tileRows = t[1]
		this.haloRows = options.inner ? (nranks > 1 ? haloRows : 0) : this.inner.haloRows
		if (!options.comm) addon.commInit(this.inner._h, uniqueId, rank, nranks)
		this.gatherHistoryRGB = nranks > 1
		this.overlapHistoryGather = nranks > 1
		// "all" (default): the whole-frame all-gather after K4, under the next frame's trace.  "bounded": no all-gather; between a frame's
		// trace and its shade only the rows of the composed GI that the tiles' rays will read travel (rfx_gather_history_rows) — fewer
		// bytes (scene dependent: 67-71 % at N = 4 / 8 on the synthetic orbit at 4K), but on the critical path (rfx_amd/tiling.py).
		this.historyGather = nranks > 1 ? (options.historyGather || "all") : "all"
		if (this.historyGather !== "bounded" && this.historyGather !== "all") throw new RangeError("historyGather: \"bounded\" or \"all\" (\"peer\": usePeerHistory)")
		this.historyBytesReceived = []
		this._haloPending = false
		this._gatherPending = false
		this.exchangeCount = 0
		// everything that is not intercepted below goes to the tile's renderer
		return new Proxy(this, {
			get: (target, prop) => {
				if (prop in target) return target[prop]
				const v = target.inner[prop]
				return typeof v === "function" ? v.bind(target.inner) : v
			}
		})
	}
	_up() {
		return this.rank + 1 < this.nranks ? this.rank + 1 : -1 // owns the rows above this tile
	}
	_down() {
		return this.rank > 0 ? this.rank - 1 : -1
	}
	// Switch the composed-GI exchange to the device-driven pull (include/rfx.h rfx_peer_*; rfx_amd/tiling.py use_peer_history).
	// `allGather(Buffer) -> [every rank's Buffer, in rank order]`: how the ranks' export blobs travel, once, by any channel the host has
	usePeerHistory(allGather) {
		if (this.nranks === 1) return
		const blobs = allGather(this._comm.peerExport(TEX.COMPOSE_RGB))
		if (!Array.isArray(blobs) || blobs.length !== this.nranks) throw new RangeError("usePeerHistory: allGather returns one blob per rank, in rank order")
		this._comm.peerOpen(TEX.COMPOSE_RGB, Buffer.concat(blobs), this.rank, this.nranks)
		this.historyGather = "peer"
	}
	exchange(texs) {
		if (this.nranks === 1 || this.haloRows === 0) return
		for (const tex of texs) this._comm.haloExchange(tex, this._up(), this._down())
		this._haloPending = true
		this.exchangeCount++
	}
	commWait() {
		if (this._haloPending || this._gatherPending) this._comm.commWait()
		this._haloPending = false
		this._gatherPending = false
	}
	// hooks called by effects.js
	afterTemporalPass() {
		this.exchange([TEX.TEMPORAL0, TEX.TEMPORAL1])
	}
	afterDenoisePass(i, uniforms) {
		this.exchange(uniforms.writeToB ? [TEX.DENOISE_B0, TEX.DENOISE_B1] : [TEX.DENOISE_A0, TEX.DENOISE_A1])
	}
	afterComposePass() {
		if (this.nranks > 1 && this.historyGather === "all") {
			this._comm.allgatherHistory(TEX.COMPOSE_RGB)
			this._gatherPending = true
		}
	}
	afterCopyFramebuffer(tex) {
		this.exchange([tex])
	}
	beforeSsgiShade() {
		if (this.nranks > 1 && this.historyGather === "bounded") {
			this.historyBytesReceived.push(this._comm.gatherHistoryRows(TEX.COMPOSE_RGB))
			this._gatherPending = true
		}
		this.commWait()
		if (this.nranks > 1 && this.historyGather === "peer") {
			this.historyBytesReceived.push(this._comm.peerGatherHistory(TEX.COMPOSE_RGB)) // (what the previous frame's pull moved)
			this._comm.commWait() // orders the shade after the pull (stream order: nothing waits on the host)
		}
	}
	ssgiMarch(u) {
		if (this.nranks > 1 && this.historyGather !== "all") throw new Error("TiledRenderer (historyGather \"" + this.historyGather + "\"): K1 must run as ssgiTrace / ssgiShade")
		this.commWait()
		this.inner.ssgiMarch(u)
	}
	temporalReproject(u) {
		this.commWait() // K2 gathers its history (exchanged at the end of the previous frame) anywhere within the halo
		this.inner.temporalReproject(u)
	}
	copyFramebuffer(dst) {
		this.commWait()
		this.inner.copyFramebuffer(dst)
	}
	poissonDenoise(u) {
		this._interiorFirst(() => this.inner.poissonDenoise(u))
	}
	compose(u) {
		this._interiorFirst(() => this.inner.compose(u))
	}
	finalCompose(u) {
		this.commWait()
		this.inner.finalCompose(u)
	}
	_interiorFirst(draw) {
		if (!this._haloPending) return draw()
		const y0 = this.tileY0, y1 = this.tileY0 + this.tileRows, h = this.haloRows
		const lo = y0 + (this.rank > 0 ? h : 0), hi = y1 - (this.rank < this.nranks - 1 ? h : 0)
		if (hi <= lo) {
			this.commWait()
			return draw()
		}
		try {
			this.inner.setRowWindow(lo, hi)
			draw()
			this.commWait()
			if (lo > y0) {
				this.inner.setRowWindow(y0, lo)
				draw()
			}
			if (hi < y1) {
				this.inner.setRowWindow(hi, y1)
				draw()
			}
		} finally {
			this.inner.setRowWindow(0, 0)
		}
	}
	download(tex, row0, rows) {
		this.commWait()
		return this.inner.download(tex, row0, rows)
	}
	sync() {
		this.commWait()
		this.inner.sync()
	}
}

module.exports = { TiledRenderer, splitRows, requiredHalo, commUniqueId: addon.commUniqueId }
