"use strict"
// Public surface of the Node host: what `import { SSGIEffect, TRAAEffect, VelocityDepthNormalPass } from "realism-effects"`
// gives for the hot path (src/index.js:1-31), plus the device (Renderer) and the dump reader.
module.exports = Object.assign({}, require("./effects"), require("./Renderer"), require("./dump"), require("./envmap"), require("./tiling"), require("./imageio"))
