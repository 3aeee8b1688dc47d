"use strict"
// Dump directory reader: frame.json + raw little-endian planes (row 0 = bottom).
//   frame.json : { width, height, camera: {...16-number matrices..., position, quaternion, near, far}, prevCamera: {...} }
//   depth.bin    Float32 W*H        gbuffer.bin  Uint32 W*H*4 (bit patterns of the RGBA32F texels)
//   velocity.bin Uint32 W*H*4       direct.bin   Float32 W*H*4
const fs = require("fs")
const path = require("path")

function plane(file, Ctor) {
	const b = fs.readFileSync(file)
	const ab = b.buffer.slice(b.byteOffset, b.byteOffset + b.length) // own, aligned ArrayBuffer
	return new Ctor(ab)
}

function readDump(dir) {
	const meta = JSON.parse(fs.readFileSync(path.join(dir, "frame.json"), "utf8"))
	const n = meta.width * meta.height
	const frame = {
		width: meta.width,
		height: meta.height,
		camera: meta.camera,
		prevCamera: meta.prevCamera,
		depth: plane(path.join(dir, "depth.bin"), Float32Array),
		gbuffer: plane(path.join(dir, "gbuffer.bin"), Uint32Array),
		velocity: plane(path.join(dir, "velocity.bin"), Uint32Array),
		direct: plane(path.join(dir, "direct.bin"), Float32Array)
	}
	if (frame.depth.length !== n || frame.gbuffer.length !== 4 * n || frame.velocity.length !== 4 * n || frame.direct.length !== 4 * n)
		throw new Error("dump " + dir + ": plane sizes do not match frame.json")
	return frame
}

module.exports = { readDump }
