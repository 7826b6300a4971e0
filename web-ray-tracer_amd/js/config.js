'use strict';
// Config of the FlexLight API (reference modules/config.js:3-16), same field names and defaults; the
// renderer re-reads it every frame.  The HIP back-end has no post-AA (FXAA/TAA are out of scope), so
// `antialiasing` defaults to off; `temporal` defaults to off as well so that a single renderFrame()
// is the plain path-traced frame (the reference's defaults are true / 'fxaa').
class Config {
  constructor () {
    this.samplesPerRay = 1;
    this.renderQuality = 1;
    this.maxReflections = 5;
    this.minImportancy = 0.3;
    this.firstPasses = 3;
    this.secondPasses = 3;
    this.temporal = false;
    this.temporalSamples = 4;
    this.filter = false;
    this.hdr = true;
    this.antialiasing = undefined;
  }
}
module.exports = { Config };
