"use strict"
// Host side of the environment map's importance sampling: what the reference computes on the CPU (in a Web Worker) for
// `sampleEquirectProbability` — src/ssgi/utils/EquirectHdrInfoUniform.js:148-245 — two inverse-CDF tables over the map's luminance and the
// luminance sum.  Scalars are JS doubles, table entries Float32Array elements, exactly as there; this is the Node twin of
// rfx_amd/envmap.py (both are pinned to the tables the reference's own code produces, tests/golden/chain_envmis_*.npz).

// first index in [from, from + count) whose value is not below `target`, capped at the last element; returned relative to `from`
function lowerBound(values, target, from, count) {
	let lo = from
	let hi = from + count - 1
	while (lo < hi) {
		const mid = (lo + hi) >> 1
		if (values[mid] < target) lo = mid + 1
		else hi = mid
	}
	return lo - from
}

// data: Float32Array(height * width * 4) texels in the order the reference's DataTexture holds them.  flipY reproduces the worker's
// "un-flip": row y is copied onto row height-1-y while y walks upwards, so the upper half is overwritten before it is read.
function buildImportance(data, width, height, flipY) {
	let texels = data
	if (flipY) {
		texels = new Float32Array(data)
		const rowLen = width * 4
		for (let y = 0; y < height; y++) texels.copyWithin((height - 1 - y) * rowLen, y * rowLen, (y + 1) * rowLen)
	}
	const rowCdf = new Float32Array(width * height)
	const colCdf = new Float32Array(height)
	const rowSum = new Float64Array(height)
	let total = 0
	let below = 0
	for (let y = 0; y < height; y++) {
		let run = 0
		for (let x = 0; x < width; x++) {
			const t = 4 * (y * width + x)
			const lum = 0.2126 * texels[t] + 0.7152 * texels[t + 1] + 0.0722 * texels[t + 2]
			run += lum
			total += lum
			rowCdf[y * width + x] = run
		}
		if (run !== 0) for (let x = 0; x < width; x++) rowCdf[y * width + x] /= run
		rowSum[y] = run
		below += run
		colCdf[y] = below
	}
	if (below !== 0) for (let y = 0; y < height; y++) colCdf[y] /= below
	// inverse CDFs at the texel centres
	const marginalWeights = new Float32Array(height)
	for (let i = 0; i < height; i++) marginalWeights[i] = (lowerBound(colCdf, (i + 1) / height, 0, height) + 0.5) / height
	const conditionalWeights = new Float32Array(width * height)
	for (let y = 0; y < height; y++)
		for (let x = 0; x < width; x++) conditionalWeights[y * width + x] = (lowerBound(rowCdf, (x + 1) / width, y * width, width) + 0.5) / width
	return { marginalWeights, conditionalWeights, totalSumValue: total }
}

module.exports = { buildImportance }
